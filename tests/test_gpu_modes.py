"""The engine has two routing kernels behind one C ABI: the per-tick streaming kernel (k_tick) and the time-tiled
kernel over subtree tiles (k_tile, default) in several shapes (positions per thread, ticks per task, tile capacity).
All must agree with the oracle; the choice is an RR_WAVE* / RR_TILE_BLOCK environment knob read at plan creation."""
import os

import numpy as np
import pytest

from conftest import assert_close
from oracle import oracle
from river_route_amd import synth
from river_route_amd.engine import DeviceBuffer, Plan

pytestmark = pytest.mark.gpu


def csc_from_down(down_index):
    has = down_index >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
    return indptr, down_index[has].astype(np.int32)


KNOBS = ('RR_WAVE', 'RR_WAVE_K', 'RR_TILE_BLOCK')
# tile capacity 64 / 333 on these networks: hundreds of tiles, 10-30 tile levels, thousands of ghosts
SHAPES = [{'RR_WAVE': '0'}, {'RR_WAVE': '1', 'RR_WAVE_K': '16'}, {'RR_WAVE': '1', 'RR_WAVE_K': '32'},
          {'RR_WAVE': '1', 'RR_TILE_BLOCK': '64', 'RR_WAVE_K': '16'},
          {'RR_WAVE': '1', 'RR_TILE_BLOCK': '333', 'RR_WAVE_K': '64'}, {'RR_WAVE': '1', 'RR_WAVE_K': '128'},
          {'RR_WAVE': '1', 'RR_WAVE_K': '256', 'RR_TILE_BLOCK': '200'}]


@pytest.mark.parametrize('env', SHAPES)
@pytest.mark.parametrize('n,T,nsub,has_lateral', [(60000, 70, 1, True), (60000, 23, 3, True), (60000, 31, 2, False)])
def test_every_kernel_shape_matches_the_oracle(monkeypatch, env, n, T, nsub, has_lateral):
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    net = synth.synth_network(n, seed=21)
    indptr, indices = csc_from_down(net.down_index)
    dt = 900.0
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, dt)
    lhs = -c1[indices]
    q0 = 5.0 * synth.u01(99, np.arange(n))
    with Plan(indptr, indices) as plan:
        if has_lateral:
            c4_dt = (c1 + c2) / (dt * nsub)
            ql = synth.synth_qlateral(n, 0, T, dt=dt * nsub)
            plan.set_coeffs(lhs, c2, c3, c4_dt)
            q_ref, d_ref = q0.copy(), np.zeros((T, n))
            oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, ql, d_ref, nsub)
            q, d = q0.copy(), np.zeros((T, n))
            plan.rapid_route(q, ql, d, nsub)
            # a second call continues from the state of the first (state hand-off between files)
            q2_ref, d2_ref = q_ref.copy(), np.zeros((T, n))
            oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q2_ref, ql, d2_ref, nsub)
            q2, d2 = q.copy(), np.zeros((T, n))
            plan.rapid_route(q2, ql, d2, nsub)
            assert_close(q2, q2_ref, 'second call q_t')
            assert_close(d2, d2_ref, 'second call discharge')
        else:
            plan.set_coeffs(lhs, c2, c3, None)
            q_ref, d_ref = q0.copy(), np.zeros((T, n))
            oracle.muskingum_route(indptr, indices, lhs, c2, c3, q_ref, d_ref, T, nsub)
            q, d = q0.copy(), np.zeros((T, n))
            plan.muskingum_route(q, d, T, nsub)
        assert_close(q, q_ref, 'q_t')
        assert_close(d, d_ref, 'discharge')


def test_per_edge_weights_fall_back_to_the_streaming_kernel():
    """lhs_off_data is per CSC entry in the reference's signature; non-uniform weights into one reach are honoured."""
    n, T = 3000, 20
    net = synth.synth_network(n, seed=2)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0)
    lhs = -c1[indices] * (1.0 + 0.1 * synth.u01(5, np.arange(indices.size)))
    c4_dt = (c1 + c2) / 900.0
    ql = synth.synth_qlateral(n, 0, T)
    q_ref, d_ref = np.zeros(n), np.zeros((T, n))
    oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, ql, d_ref, 1)
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(lhs, c2, c3, c4_dt)
        q, d = np.zeros(n), np.zeros((T, n))
        plan.rapid_route(q, ql, d, 1)
    assert_close(q, q_ref, 'q_t')
    assert_close(d, d_ref, 'discharge')


@pytest.mark.parametrize('env', SHAPES + [{}])
@pytest.mark.parametrize('n,T,nsub,n_ks', [(40000, 50, 1, 48), (40000, 17, 3, 5)])
def test_unit_route_every_kernel_shape(monkeypatch, env, n, T, nsub, n_ks):
    """UnitMuskingum through the streaming kernel (k_tick_unit) and the time-tiled kernel (k_tile, UNIT) vs the oracle,
    two consecutive files with state hand-off as UnitMuskingum._router does it."""
    from conftest import unit_split
    from river_route_amd.engine import uh_convolve
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    net = synth.synth_network(n, seed=31)
    indptr, indices = csc_from_down(net.down_index)
    hw_idx, inner_idx, A_in, A_hw = unit_split(indptr, indices, n)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0 / nsub)
    c1i, c2i, c3i = c1[inner_idx], c2[inner_idx], c3[inner_idx]
    args = (A_in.indptr, A_in.indices, -c1i[A_in.indices], A_in.indptr, A_in.indices, A_in.data,
            A_hw.indptr, A_hw.indices, A_hw.data, c1i, c2i, c3i, hw_idx, inner_idx)
    kern = synth.synth_uh_kernel(n, n_ks)
    uh = oracle.UnitHydrograph(kern)
    st = np.zeros_like(kern)
    state_ref = 3.0 * synth.u01(7, np.arange(n))
    state = state_ref.copy()
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(-c1[indices], c2, c3, None)
        for f in range(2):
            depth = synth.synth_runoff_depth(n, f * T, (f + 1) * T)
            conv_ref = uh.convolve(depth)
            conv = uh_convolve(kern, st, depth)
            assert_close(conv, conv_ref, f'file {f} convolved')
            qc_ref, qf_ref, d_ref = state_ref[inner_idx].copy(), state_ref[inner_idx].copy(), np.zeros((T, n))
            oracle.unit_route(*args, qc_ref, qf_ref, conv_ref, d_ref, nsub)
            state_ref[hw_idx], state_ref[inner_idx] = conv_ref[-1][hw_idx], qf_ref
            qc, qf, d = state[inner_idx].copy(), state[inner_idx].copy(), np.zeros((T, n))
            plan.unit_route(qc, qf, conv_ref, d, nsub)
            state[hw_idx], state[inner_idx] = conv_ref[-1][hw_idx], qf
            assert_close(qc, qc_ref, f'file {f} q_ch')
            assert_close(qf, qf_ref, f'file {f} q_full')
            assert_close(d, d_ref, f'file {f} discharge')
            np.testing.assert_array_equal(d[:, hw_idx], conv_ref[:, hw_idx])


def _route_vs_oracle(down, T=25, nsub=1, seed=3):
    n = down.shape[0]
    indptr, indices = csc_from_down(down)
    k = 900.0 + 6300.0 * synth.u01(seed, np.arange(n))
    x = 0.05 + 0.4 * synth.u01(seed + 1, np.arange(n))
    c1, c2, c3 = oracle.muskingum_coefficients(k, x, 900.0)
    lhs = -c1[indices]
    c4_dt = (c1 + c2) / (900.0 * nsub)
    ql = synth.synth_qlateral(n, 0, T, dt=900.0 * nsub)
    q0 = 2.0 * synth.u01(seed + 2, np.arange(n))
    q_ref, d_ref = q0.copy(), np.zeros((T, n))
    oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, ql, d_ref, nsub)
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(lhs, c2, c3, c4_dt)
        q, d = q0.copy(), np.zeros((T, n))
        plan.rapid_route(q, ql, d, nsub)
        depth, widest = plan.depth, plan.widest_level
    assert_close(q, q_ref, 'q_t')
    assert_close(d, d_ref, 'discharge')
    return depth, widest


@pytest.mark.parametrize('wave', ['0', '1'])
def test_degenerate_network_shapes(monkeypatch, wave):
    """Extremes of the lag pipeline: a single chain (depth = n, every level one reach wide), a star (one level of
    5,000 tributaries into one outlet: in-degree 5,000, more than a tile holds, so it streams),
    unconnected reaches only, and a comb (a main stem with one tributary per reach)."""
    monkeypatch.setenv('RR_WAVE', wave)
    n = 3000
    chain = np.arange(1, n + 1, dtype=np.int64)
    chain[-1] = -1
    assert _route_vs_oracle(chain, T=12) == (n, 1)
    star = np.full(5001, 5000, dtype=np.int64)
    star[-1] = -1
    assert _route_vs_oracle(star, T=9, nsub=2) == (2, 5000)
    lone = np.full(777, -1, dtype=np.int64)
    assert _route_vs_oracle(lone, T=5) == (1, 777)
    m = 1500   # comb: reaches 0..m-1 are tributaries, m..2m-1 the stem
    comb = np.concatenate([m + np.arange(m), m + 1 + np.arange(m)]).astype(np.int64)
    comb[-1] = -1
    assert _route_vs_oracle(comb, T=14)[0] == m + 1


def _device_route(plan, q0, ql, T, nsub, out_rows=None):
    from river_route_amd.engine import DeviceBuffer
    n = q0.shape[0]
    out_rows = out_rows or T
    d_q = DeviceBuffer(n * 8).upload(q0)
    d_ql = DeviceBuffer(ql.nbytes).upload(ql)
    d_out = DeviceBuffer(out_rows * n * 8)
    plan.rapid_route_dev(d_q, d_ql, ql.shape[0], d_out, out_rows, T, nsub)
    q = d_q.download(np.float64, (n,))
    d = d_out.download(np.float64, (out_rows, n))
    for b in (d_q, d_ql, d_out):
        b.free()
    return q, d


def test_degenerate_network_shapes_record_mode(monkeypatch):
    """The same extremes through the device-resident entry point with the time-tiled kernel forced: a confluence of
    3,000 tributaries (more upstream reaches than a tile holds: not tileable, streams), a comb (every stem reach has a
    ghost-free tributary), a chain (every tile of the skeleton one level above the last) and unconnected reaches."""
    monkeypatch.setenv('RR_WAVE', '1')
    for k in KNOBS[1:]:
        monkeypatch.delenv(k, raising=False)
    m = 3000
    fan = np.concatenate([np.full(m, m), m + 1 + np.arange(60)]).astype(np.int64)      # m tributaries -> reach m -> chain
    fan[-1] = -1
    mm = 1500
    comb = np.concatenate([mm + np.arange(mm), mm + 1 + np.arange(mm)]).astype(np.int64)
    comb[-1] = -1
    chain = np.arange(1, 2501, dtype=np.int64)
    chain[-1] = -1
    lone = np.full(777, -1, dtype=np.int64)
    for down, T in ((fan, 40), (comb, 33), (chain, 20), (lone, 17)):
        n = down.shape[0]
        indptr, indices = csc_from_down(down)
        k_, x_ = 900.0 + 6300.0 * synth.u01(5, np.arange(n)), 0.05 + 0.4 * synth.u01(6, np.arange(n))
        c1, c2, c3 = oracle.muskingum_coefficients(k_, x_, 900.0)
        lhs, c4_dt = -c1[indices], (c1 + c2) / 900.0
        ql = synth.synth_qlateral(n, 0, T)
        q0 = 2.0 * synth.u01(7, np.arange(n))
        q_ref, d_ref = q0.copy(), np.zeros((T, n))
        oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, ql, d_ref, 1)
        with Plan(indptr, indices) as plan:
            plan.set_coeffs(lhs, c2, c3, c4_dt)
            q, d = _device_route(plan, q0, ql, T, 1)
        assert_close(q, q_ref, f'q_t n={n}')
        assert_close(d, d_ref, f'discharge n={n}')


@pytest.mark.parametrize('env', [{'RR_WAVE': '1'}, {}, {'RR_WAVE': '0'}, {'RR_WAVE': '1', 'RR_WAVE_K': '16', 'RR_TILE_BLOCK': '777'}])
@pytest.mark.parametrize('n,T,ql_rows', [(60000, 100, 100), (60000, 7, 7), (3000, 6000, 48), (60000, 5000, 96),
                                        (300000, 70, 70),      # > 256k reaches: the two-positions-per-thread shapes
                                        (1000000, 80, 80)])    # BASELINE size: the bench's kernel shapes, full oracle comparison
def test_device_resident_route_record_mode(monkeypatch, env, n, T, ql_rows):
    """Device arrays in params order (the bench path): record ring + one-pass permutation around the time-tiled kernel
    (default) and the streaming kernel, incl. cyclic forcing and ring wrap-around."""
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    net = synth.synth_network(n, seed=13)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0)
    lhs, c4_dt = -c1[indices], (c1 + c2) / 900.0
    ql = synth.synth_qlateral(n, 0, ql_rows)
    q0 = 4.0 * synth.u01(1, np.arange(n))
    ql_full = np.ascontiguousarray(np.tile(ql, (T // ql_rows + 1, 1))[:T])
    q_ref, d_ref = q0.copy(), np.zeros((T, n))
    oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, ql_full, d_ref, 1)
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(lhs, c2, c3, c4_dt)
        q, d = _device_route(plan, q0, ql, T, 1)
        assert_close(q, q_ref, 'q_t')
        assert_close(d, d_ref, 'discharge')
        # again on the same plan, into a cyclic sink: only the last rows survive
        sink = min(T, 32)
        q2, d2 = _device_route(plan, q0, ql, T, 1, out_rows=sink)
        np.testing.assert_array_equal(q2, q)
        last = np.arange(T - sink, T)
        np.testing.assert_array_equal(d2[last % sink], d[last])


@pytest.mark.parametrize('mode,nsub', [('rapid', 3), ('muskingum', 2), ('unit', 1), ('unit', 2)])
def test_record_ring_goes_round_with_substeps_and_other_modes(monkeypatch, mode, nsub):
    """The record ring recycling its slots (a call several times longer than the ring) with sub-steps, without lateral rows
    (no in-pass: the tasks write fresh records) and for UnitMuskingum -- each against the oracle, every row."""
    from conftest import unit_split
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv('RR_WAVE', '1')
    monkeypatch.setenv('RR_VERBOSE', '1')
    n, T, dt = 6000, 2400 // nsub, 900.0 / nsub
    net = synth.synth_network(n, seed=77)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, dt)
    q0 = 3.0 * synth.u01(5, np.arange(n))
    from river_route_amd.engine import DeviceBuffer
    with Plan(indptr, indices) as plan:
        if mode == 'unit':
            hw_idx, inner_idx, A_in, A_hw = unit_split(indptr, indices, n)
            c1i, c2i, c3i = c1[inner_idx], c2[inner_idx], c3[inner_idx]
            args = (A_in.indptr, A_in.indices, -c1i[A_in.indices], A_in.indptr, A_in.indices, A_in.data,
                    A_hw.indptr, A_hw.indices, A_hw.data, c1i, c2i, c3i, hw_idx, inner_idx)
            conv = synth.synth_qlateral(n, 0, T) / 900.0 - 0.2       # some negative lateral: the clip at zero has work to do
            qc_ref, qf_ref, d_ref = 0.5 * q0[inner_idx], q0[inner_idx].copy(), np.zeros((T, n))
            oracle.unit_route(*args, qc_ref, qf_ref, conv, d_ref, nsub)
            plan.set_coeffs(-c1[indices], c2, c3, None)
            ni = inner_idx.size
            d_qc, d_qf = DeviceBuffer(ni * 8).upload(0.5 * q0[inner_idx]), DeviceBuffer(ni * 8).upload(q0[inner_idx].copy())
            d_conv, d_out = DeviceBuffer(conv.nbytes).upload(conv), DeviceBuffer(T * n * 8)
            plan.unit_route_dev(d_qc, d_qf, d_conv, T, d_out, T, T, nsub)
            assert_close(d_out.download(np.float64, (T, n)), d_ref, 'discharge')
            assert_close(d_qc.download(np.float64, (ni,)), qc_ref, 'q_ch')
            assert_close(d_qf.download(np.float64, (ni,)), qf_ref, 'q_full')
            np.testing.assert_array_equal(d_out.download(np.float64, (T, n))[:, hw_idx], conv[:, hw_idx])      # headwaters: not clipped
            for b in (d_qc, d_qf, d_conv, d_out):
                b.free()
        elif mode == 'rapid':
            c4_dt = (c1 + c2) / dt
            ql = synth.synth_qlateral(n, 0, T) - 150.0               # some negative lateral
            q_ref, d_ref = q0.copy(), np.zeros((T, n))
            oracle.rapid_route(indptr, indices, -c1[indices], c2, c3, c4_dt, q_ref, ql, d_ref, nsub)
            plan.set_coeffs(-c1[indices], c2, c3, c4_dt)
            q, d = _device_route(plan, q0, ql, T, nsub)
            assert (d_ref == 0.0).any()
            assert_close(q, q_ref, 'q_t')
            assert_close(d, d_ref, 'discharge')
        else:
            q_ref, d_ref = q0.copy(), np.zeros((T, n))
            oracle.muskingum_route(indptr, indices, -c1[indices], c2, c3, q_ref, d_ref, T, nsub)
            plan.set_coeffs(-c1[indices], c2, c3, None)
            d_q, d_out = DeviceBuffer(n * 8).upload(q0), DeviceBuffer(T * n * 8)
            plan.muskingum_route_dev(d_q, d_out, T, T, nsub)
            assert_close(d_q.download(np.float64, (n,)), q_ref, 'q_t')
            assert_close(d_out.download(np.float64, (T, n)), d_ref, 'discharge')
            d_q.free(); d_out.free()
        assert plan.profile()['ticks_per_launch'] >= 16


@pytest.mark.parametrize('env', [{'RR_WAVE': '0'}, {}])
def test_dev_entry_points_only_enqueue(monkeypatch, env):
    """include/rr_hip.h: a *_dev call allocates nothing.  Through the raw ABI: without rr_plan_reserve the call is refused with
    RR_E_STATE and a message that names the remedy; after it, the call leaves the device's free memory where it was and
    agrees with the oracle; a larger call than reserved is refused again, a smaller one fits."""
    import ctypes as C
    import torch
    from river_route_amd import _lib
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n, T, nsub = 50_000, 96, 1
    net = synth.synth_network(n, seed=5)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0)
    lhs, c4_dt = -c1[indices], (c1 + c2) / 900.0
    ql = synth.synth_qlateral(n, 0, T)
    q_ref, d_ref = np.zeros(n), np.zeros((T, n))
    oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, ql, d_ref, nsub)
    L = _lib.lib()
    h = C.c_void_p()
    assert L.rr_plan_create(n, _lib.ptr(indptr), _lib.ptr(indices), 0, C.byref(h)) == 0
    try:
        assert L.rr_plan_set_coeffs(h, _lib.ptr(lhs), _lib.ptr(c2), _lib.ptr(c3), _lib.ptr(c4_dt)) == 0
        dev = torch.device('cuda', 0)
        d_q = torch.zeros(n, dtype=torch.float64, device=dev)
        d_ql = torch.from_numpy(ql).to(dev)
        d_out = torch.zeros((2 * T, n), dtype=torch.float64, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        call = lambda rows: L.rr_rapid_route_dev(h, _lib.ptr(d_q), _lib.ptr(d_ql), T, _lib.ptr(d_out), 2 * T, rows, nsub, stream)  # noqa: E731
        assert call(T) == _lib.RR_E_STATE and b'rr_plan_reserve' in L.rr_last_error()
        info = np.zeros(8, dtype=np.int64)
        assert L.rr_plan_reserve(h, 0, T, nsub, 0, _lib.ptr(info)) == 0
        assert info[0] == (0 if env else 1) and info[3] >= info[7] > 0
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info(0)[0]
        assert call(T) == 0
        free1 = torch.cuda.mem_get_info(0)[0]      # asked while the call may still be running: nothing in it synchronised either
        torch.cuda.synchronize()
        assert free0 == free1 == torch.cuda.mem_get_info(0)[0]
        assert_close(d_out[:T].cpu().numpy(), d_ref, 'discharge')
        assert_close(d_q.cpu().numpy(), q_ref, 'q_t')
        assert call(T // 2) == 0                                   # a smaller call fits
        assert call(64 * T) == _lib.RR_E_STATE                     # a larger one is refused, not grown into
        assert L.rr_plan_reserve(h, 0, 64 * T, nsub, 0, None) == 0 and call(64 * T) == 0
        torch.cuda.synchronize()
    finally:
        L.rr_plan_destroy(h)


def test_profile_names_every_kernel_of_the_path(monkeypatch):
    """bench.py's whole-path block: with sampling on, rr_plan_profile / rr_plan_profile_aux time the routing kernel and the two
    record passes of the last call between HIP events on its stream, and rr_plan_last_kernel says which routing kernel ran."""
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    n, T = 60_000, 1500
    net = synth.synth_network(n, seed=5)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0)
    ql = synth.synth_qlateral(n, 0, 64)
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(-c1[indices], c2, c3, (c1 + c2) / 900.0)
        plan.set_options(sample_every=128)
        d_q, d_ql, d_out = DeviceBuffer(n * 8).upload(np.zeros(n)), DeviceBuffer(ql.nbytes).upload(ql), DeviceBuffer(128 * n * 8)
        plan.rapid_route_dev(d_q, d_ql, 64, d_out, 128, T, 1)
        prof, aux = plan.profile(), plan.profile_aux()
        assert plan.last_kernel() == 'tile' and prof['brackets'] > 0 and prof['sampled_ms'] > 0
        batches = (T + 127) // 128
        assert aux['k_rec_in']['launches'] == (T + 14) // 128 + 1 and aux['k_rec_out']['launches'] == batches
        for k in ('k_rec_in', 'k_rec_out'):
            assert aux[k]['sampled'] == (aux[k]['launches'] + 3) // 4 and 0 < aux[k]['sampled_ms'] < 1e3
        for b in (d_q, d_ql, d_out):
            b.free()


@pytest.mark.gpu
def test_rows_between_a_file_and_the_device(tmp_path):
    """rr_rows_upload / rr_rows_download (include/rr_hip.h): rows that lie `pitch` bytes apart in a file <-> device rows, several staging
    chunks long; a file that ends before the last row, a file that does not exist and bad arguments are refused with an error, not a crash."""
    from river_route_amd import engine
    from river_route_amd._lib import RRError
    rows, cols = 700, 40_000      # 112 MB: two 64 MB staging chunks
    rng = np.random.default_rng(3)
    data = rng.standard_normal((rows, cols)).astype(np.float32)
    pad, head = 24, 128      # bytes between rows (a record variable's other members) and in front of the first row
    path = tmp_path / 'rows.bin'
    raw = np.zeros((rows, cols * 4 + pad), dtype=np.uint8)
    raw[:, :cols * 4] = data.view(np.uint8).reshape(rows, cols * 4)
    with open(path, 'wb') as f:
        f.write(b'\0' * head)
        f.write(raw.tobytes())
    dev = DeviceBuffer(rows * cols * 4)
    engine.rows_upload(dev, cols * 4, path, head, cols * 4 + pad, cols * 4, rows)
    np.testing.assert_array_equal(dev.download(np.float32, (rows, cols)), data)
    out = tmp_path / 'out.bin'
    with open(out, 'wb') as f:
        f.truncate(head + rows * (cols * 4 + pad))
    engine.rows_download(dev, cols * 4, out, head, cols * 4 + pad, cols * 4, rows)
    got = np.fromfile(out, dtype=np.uint8)[head:].reshape(rows, cols * 4 + pad)
    np.testing.assert_array_equal(got[:, :cols * 4].copy().view(np.float32).reshape(rows, cols), data)
    assert not got[:, cols * 4:].any(), 'the bytes between the rows were left alone'
    with pytest.raises(RRError, match='short read'):
        engine.rows_upload(dev, cols * 4, path, head, cols * 4 + pad, cols * 4, rows + 5)
    with pytest.raises(RRError, match='cannot open'):
        engine.rows_upload(dev, cols * 4, tmp_path / 'missing.bin', 0, cols * 4, cols * 4, rows)
    with pytest.raises(RRError, match='cannot open'):
        engine.rows_download(dev, cols * 4, tmp_path / 'missing_out.bin', 0, cols * 4, cols * 4, rows)
    with pytest.raises(RRError):
        engine.rows_upload(dev, cols * 4 - 4, path, head, cols * 4 + pad, cols * 4, rows)      # device pitch shorter than a row
    np.testing.assert_array_equal(dev.download(np.float32, (rows, cols))[:rows - 5], data[:rows - 5])      # (the failed calls left the device usable)
    dev.free()

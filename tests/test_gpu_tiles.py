"""The time-tiled kernel (k_tile) on exactly the paths bench.py times and the routers use: device arrays, forced
(RR_WAVE=1), BASELINE sizes, every mode (Rapid / Muskingum / Unit), sub-steps, consecutive files -- each against the
oracle -- and the partitioned path against the oracle at the part size of BASELINE config 5."""
import numpy as np
import pytest

from conftest import assert_close, unit_split
from oracle import oracle
from river_route_amd import synth
from river_route_amd.engine import DeviceBuffer, Plan, partition_forest, uh_convolve

pytestmark = pytest.mark.gpu
KNOBS = ('RR_WAVE', 'RR_WAVE_K', 'RR_TILE_BLOCK', 'RR_TILE_LEAN', 'RR_UH_PAIRS', 'RR_DIRECT')


def csc_from_down(down_index):
    has = down_index >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
    return indptr, down_index[has].astype(np.int32)


def set_env(monkeypatch, env):
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)


@pytest.mark.parametrize('n,T,env,wide', [(300_000, 80, {'RR_WAVE': '1'}, False), (1_000_000, 80, {'RR_WAVE': '1'}, False),
                                           (300_000, 70, {'RR_WAVE': '1', 'RR_WAVE_K': '32'}, False),
                                           (60_000, 50, {'RR_WAVE': '1', 'RR_TILE_BLOCK': '64', 'RR_WAVE_K': '16'}, False),
                                           (120_000, 70, {'RR_WAVE': '1', 'RR_TILE_LEAN': '0'}, False),      # the general tick for every tile
                                           (120_000, 70, {'RR_WAVE': '1'}, True)])      # confluences of four to eight reaches: their tiles go to the general kernel beside the short tick
def test_unit_route_dev_time_tiled_vs_oracle(monkeypatch, n, T, env, wide):
    """BASELINE config 4's timed kernel (k_tile, UNIT; the short tick unless RR_TILE_LEAN=0) over many tiles and tile levels,
    device arrays, two consecutive files with the state hand-off of UnitMuskingum._router (river_route/routers/UnitMuskingum.py:72-98)."""
    set_env(monkeypatch, env)
    net = synth.synth_network(n, seed=31)
    down = _wide_network(n) if wide else net.down_index
    indptr, indices = csc_from_down(down)
    hw_idx, inner_idx, A_in, A_hw = unit_split(indptr, indices, n)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0)
    c1i, c2i, c3i = c1[inner_idx], c2[inner_idx], c3[inner_idx]
    args = (A_in.indptr, A_in.indices, -c1i[A_in.indices], A_in.indptr, A_in.indices, A_in.data,
            A_hw.indptr, A_hw.indices, A_hw.data, c1i, c2i, c3i, hw_idx, inner_idx)
    n_ks = 12
    kern = synth.synth_uh_kernel(n, n_ks)
    uh = oracle.UnitHydrograph(kern)
    state_ref = 3.0 * synth.u01(7, np.arange(n))
    state = state_ref.copy()
    ni = inner_idx.size
    with Plan(indptr, indices) as plan:
        assert plan.tile_info()['ok'] and plan.tile_info()['levels'] > 2
        plan.set_coeffs(-c1[indices], c2, c3, None)
        d_conv, d_out = DeviceBuffer(T * n * 8), DeviceBuffer(T * n * 8)
        d_qc, d_qf = DeviceBuffer(ni * 8), DeviceBuffer(ni * 8)
        for f in range(2):
            conv_ref = uh.convolve(synth.synth_runoff_depth(n, f * T, (f + 1) * T))
            qc_ref, qf_ref, d_ref = state_ref[inner_idx].copy(), state_ref[inner_idx].copy(), np.zeros((T, n))
            oracle.unit_route(*args, qc_ref, qf_ref, conv_ref, d_ref, 1)
            state_ref[hw_idx], state_ref[inner_idx] = conv_ref[-1][hw_idx], qf_ref
            d_conv.upload(conv_ref)
            d_qc.upload(state[inner_idx].copy()); d_qf.upload(state[inner_idx].copy())
            plan.unit_route_dev(d_qc, d_qf, d_conv, T, d_out, T, T, 1)
            qc, qf, d = d_qc.download(np.float64, (ni,)), d_qf.download(np.float64, (ni,)), d_out.download(np.float64, (T, n))
            state[hw_idx], state[inner_idx] = conv_ref[-1][hw_idx], qf
            assert_close(qc, qc_ref, f'file {f} q_ch')
            assert_close(qf, qf_ref, f'file {f} q_full')
            assert_close(d, d_ref, f'file {f} discharge')
            np.testing.assert_array_equal(d[:, hw_idx], conv_ref[:, hw_idx])     # headwaters: unclamped, un-averaged
        for b in (d_conv, d_out, d_qc, d_qf):
            b.free()


@pytest.mark.parametrize('n,T,nsub,mode', [(1_000_000, 40, 4, 'rapid'), (1_000_000, 36, 3, 'rapid'), (300_000, 30, 4, 'muskingum'),
                                           (300_000, 40, 3, 'unit')])
def test_substeps_time_tiled_vs_oracle(monkeypatch, n, T, nsub, mode):
    """num_substeps > 1 (river_route/routers/_numba_kernels.py:66-84): records in sub-step space, the row mean in the slot
    of the row's last sub-step, two consecutive calls."""
    set_env(monkeypatch, {'RR_WAVE': '1'})
    net = synth.synth_network(n, seed=5)
    indptr, indices = csc_from_down(net.down_index)
    dt = 900.0
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, dt)
    lhs = -c1[indices]
    q0 = 5.0 * synth.u01(99, np.arange(n))
    with Plan(indptr, indices) as plan:
        d_out = DeviceBuffer(T * n * 8)
        if mode == 'rapid':
            c4_dt = (c1 + c2) / (dt * nsub)
            plan.set_coeffs(lhs, c2, c3, c4_dt)
            d_q, d_ql = DeviceBuffer(n * 8).upload(q0), DeviceBuffer(T * n * 8)
            q_ref = q0.copy()
            for f in range(2):
                ql = synth.synth_qlateral(n, f * T, (f + 1) * T, dt=dt * nsub)
                d_ref = np.zeros((T, n))
                oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, ql, d_ref, nsub)
                d_ql.upload(ql)
                plan.rapid_route_dev(d_q, d_ql, T, d_out, T, T, nsub)
                assert_close(d_out.download(np.float64, (T, n)), d_ref, f'call {f} discharge')
                assert_close(d_q.download(np.float64, (n,)), q_ref, f'call {f} q_t')
            d_q.free(); d_ql.free()
        elif mode == 'muskingum':
            plan.set_coeffs(lhs, c2, c3, None)
            q_ref, d_ref = q0.copy(), np.zeros((T, n))
            oracle.muskingum_route(indptr, indices, lhs, c2, c3, q_ref, d_ref, T, nsub)
            d_q = DeviceBuffer(n * 8).upload(q0)
            plan.muskingum_route_dev(d_q, d_out, T, T, nsub)
            assert_close(d_out.download(np.float64, (T, n)), d_ref, 'discharge')
            assert_close(d_q.download(np.float64, (n,)), q_ref, 'q_t')
            d_q.free()
        else:
            hw_idx, inner_idx, A_in, A_hw = unit_split(indptr, indices, n)
            cs1, cs2, cs3 = oracle.muskingum_coefficients(net.k, net.x, dt / nsub)
            c1i, c2i, c3i = cs1[inner_idx], cs2[inner_idx], cs3[inner_idx]
            plan.set_coeffs(-cs1[indices], cs2, cs3, None)
            conv = synth.synth_qlateral(n, 0, T) / 900.0
            qc_ref, qf_ref, d_ref = q0[inner_idx].copy(), q0[inner_idx].copy(), np.zeros((T, n))
            oracle.unit_route(A_in.indptr, A_in.indices, -c1i[A_in.indices], A_in.indptr, A_in.indices, A_in.data,
                              A_hw.indptr, A_hw.indices, A_hw.data, c1i, c2i, c3i, hw_idx, inner_idx, qc_ref, qf_ref, conv, d_ref, nsub)
            ni = inner_idx.size
            d_qc, d_qf = DeviceBuffer(ni * 8).upload(q0[inner_idx].copy()), DeviceBuffer(ni * 8).upload(q0[inner_idx].copy())
            d_conv = DeviceBuffer(T * n * 8).upload(conv)
            plan.unit_route_dev(d_qc, d_qf, d_conv, T, d_out, T, T, nsub)
            assert_close(d_out.download(np.float64, (T, n)), d_ref, 'discharge')
            assert_close(d_qf.download(np.float64, (ni,)), qf_ref, 'q_full')
            assert_close(d_qc.download(np.float64, (ni,)), qc_ref, 'q_ch')
            for b in (d_qc, d_qf, d_conv):
                b.free()
        d_out.free()


def test_short_file_at_1m_uses_the_time_tiled_kernel(monkeypatch, capfd):
    """A one-month hourly file (744 rows) at 1M reaches: the schedule's skew is (tile levels x K) ticks, so calls of this
    length are time-tiled (they streamed in round 1), and the record ring stays a small part of the card."""
    set_env(monkeypatch, {})
    monkeypatch.setenv('RR_VERBOSE', '1')
    n, T = 1_000_000, 744
    net = synth.synth_network(n)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0)
    lhs, c4_dt = -c1[indices], (c1 + c2) / 900.0
    rows = 48
    ql = synth.synth_qlateral(n, 0, rows)
    Tc = 60           # the oracle checks the first rows of the call; the cyclic sink keeps the last
    q_ref, d_ref = np.zeros(n), np.zeros((Tc, n))
    oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, np.ascontiguousarray(np.tile(ql, (2, 1))[:Tc]), d_ref, 1)
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(lhs, c2, c3, c4_dt)
        d_q, d_ql, d_out = DeviceBuffer(n * 8).upload(np.zeros(n)), DeviceBuffer(ql.nbytes).upload(ql), DeviceBuffer(T * n * 8)
        plan.rapid_route_dev(d_q, d_ql, rows, d_out, T, T, 1)
        d = d_out.download(np.float64, (Tc, n))
        for b in (d_q, d_ql, d_out):
            b.free()
    assert_close(d, d_ref, 'first rows of the file')
    err = capfd.readouterr().err
    line = [ln for ln in err.splitlines() if ln.startswith('rr: n=1000000 T=744')][-1]
    assert 'tiled=1' in line
    ring_gb = float(line.split('(')[1].split(' GB')[0])
    assert ring_gb <= 48.0, line


def _route_parts_vs_oracle(n, parts, T, chunk, sample_cols=None, seed=4):
    from river_route_amd.multi_gpu import HipPartEngine, run_in_process, split_network
    net = synth.synth_network(n, seed=seed) if isinstance(n, int) else n
    n = net.n
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0)
    c4_dt = (c1 + c2) / 900.0
    q0 = 4.0 * synth.u01(8, np.arange(n))
    ql = synth.synth_qlateral(n, 0, T)
    q_ref, d_ref = q0.copy(), np.zeros((T, n))
    oracle.rapid_route(indptr, indices, -c1[indices], c2, c3, c4_dt, q_ref, ql, d_ref, 1)
    part_of, sizes = partition_forest(indptr, indices, parts)
    specs = [split_network(net.down_index, part_of, p, parts) for p in range(parts)]
    engines = [HipPartEngine(s, c1, c2, c3, c4_dt, q0, ql[:, s.real_global], T, 1, 0, out_rows=T) for s in specs]
    run_in_process(engines, specs, T, 1, chunk)
    q, d = np.zeros(n), np.zeros((T, n))
    for s, e in zip(specs, engines):
        q[s.real_global] = e.final_state()
        d[:, s.real_global] = e.discharge.cpu().numpy()[:, s.n_ghost:]
    assert_close(q, q_ref, 'state')
    assert_close(d, d_ref, 'discharge')
    return specs


@pytest.mark.parametrize('wave', ['0', '1'])
def test_partitioned_path_vs_oracle_eight_parts(monkeypatch, wave):
    """BASELINE config 5's path (8 parts, boundary series handed downstream in batches) against the ORACLE on the
    undivided network: the trunk partition of a random network, and the nested min-max cut a chain-like network
    falls back to."""
    set_env(monkeypatch, {'RR_WAVE': wave})
    specs = _route_parts_vs_oracle(200_000, 8, 150, 32)
    assert sum(s.n_ghost for s in specs) > 50
    # chain-like: a comb whose stem is far longer than a share
    m = 30_000
    comb = np.concatenate([m + np.arange(m), m + 1 + np.arange(m)]).astype(np.int64)
    comb[-1] = -1
    net = synth.synth_network(2 * m, seed=9)
    net.down_index[:] = comb
    net.river_ids[:] = np.arange(2 * m)
    net.downstream_ids[:] = comb
    specs = _route_parts_vs_oracle(net, 8, 70, 16)
    assert max(len(s.upstream_parts) for s in specs) >= 1


def test_trunk_part_of_config5_size_vs_oracle(monkeypatch):
    """One GPU's share of BASELINE config 5 (10M reaches / 8): the trunk part of a 5M-reach network cut in four is
    1.25M reaches deep in the main stems, with hundreds of boundary inflows.  Its ghost series come from the oracle on
    the whole network; its result is compared with the oracle's on sampled rows and all of its reaches."""
    from river_route_amd.multi_gpu import split_network
    set_env(monkeypatch, {'RR_WAVE': '1'})
    monkeypatch.setenv('RR_VERBOSE', '1')
    n, parts, T = 5_000_000, 4, 96
    net = synth.synth_network(n)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0)
    c4_dt = (c1 + c2) / 900.0
    part_of, sizes = partition_forest(indptr, indices, parts)
    spec = split_network(net.down_index, part_of, parts - 1, parts)
    assert spec.real_global.size >= 1_200_000 and spec.n_ghost > 100
    members = np.concatenate([spec.ghost_global, spec.real_global])
    # oracle on the whole network, one 16-row block at a time (a (T, 5M) array would be 3.8 GB)
    q = np.zeros(n)
    ghost_series = np.zeros((T, spec.n_ghost))
    d_ref = np.zeros((T, spec.real_global.size))
    ql_part = np.zeros((T, members.size))
    for t0 in range(0, T, 16):
        ql = synth.synth_qlateral(n, t0, t0 + 16)
        d = np.zeros((16, n))
        # export series need the unclamped state: route row by row
        for r in range(16):
            oracle.rapid_route(indptr, indices, -c1[indices], c2, c3, c4_dt, q, ql[r:r + 1], d[r:r + 1], 1)
            ghost_series[t0 + r] = q[spec.ghost_global]
        d_ref[t0:t0 + 16] = d[:, spec.real_global]
        ql_part[t0:t0 + 16, spec.n_ghost:] = ql[:, spec.real_global]
    q_ref = q[spec.real_global]
    c1m = c1[members]
    has = spec.down_local >= 0
    with Plan(spec.indptr, spec.indices) as plan:
        info = plan.tile_info()
        assert info['ok']
        def loc(v):
            a = v[members].copy(); a[:spec.n_ghost] = 0.0; return a
        plan.set_coeffs(-c1m[spec.down_local[has]], loc(c2), loc(c3), loc(c4_dt))
        export_local = spec.n_ghost + np.searchsorted(spec.real_global, spec.export_global)
        plan.set_boundary(np.arange(spec.n_ghost), export_local)
        nl = members.size
        d_q = DeviceBuffer(nl * 8).upload(np.zeros(nl))
        d_ql, d_out = DeviceBuffer(ql_part.nbytes).upload(ql_part), DeviceBuffer(T * nl * 8)
        d_g = DeviceBuffer(ghost_series.nbytes).upload(ghost_series)
        d_e = DeviceBuffer(max(1, T * max(1, spec.export_global.size)) * 8)
        plan.stream_begin(d_q, d_ql, T, d_out, T, T, 1, d_g, d_e)
        plan.stream_advance(T, T)
        plan.stream_end(d_q)
        d = d_out.download(np.float64, (T, nl))[:, spec.n_ghost:]
        qf = d_q.download(np.float64, (nl,))[spec.n_ghost:]
        for b in (d_q, d_ql, d_out, d_g, d_e):
            b.free()
    assert_close(qf, q_ref, 'state of the trunk part')
    assert_close(d, d_ref, 'discharge of the trunk part')


def test_uhkernels_convolve_equals_incremental_on_the_gpu():
    """The reference's own known-answer test (tests/test_uhkernels.py:52-78) on river_route_amd.uhkernels.UnitHydrograph:
    convolve() on the whole series == convolve_incrementally() step by step, rtol 1e-12; and the impulse response
    reproduces the kernel (tests/test_uhkernels.py:81-99)."""
    from river_route_amd.uhkernels import UnitHydrograph
    rng = np.random.default_rng(123)
    n_ks, n, T = 3, 4, 10
    kernel = rng.random((n_ks, n))
    kernel /= kernel.sum(axis=0)
    runoff = rng.random((T, n))
    whole = UnitHydrograph.from_array(kernel).convolve(runoff)
    uh = UnitHydrograph.from_array(kernel)
    stepwise = np.stack([uh.convolve_incrementally(runoff[t]) for t in range(T)])
    np.testing.assert_allclose(whole, stepwise, rtol=1e-12, atol=0)
    impulse = np.zeros((n_ks + 2, n))
    impulse[0] = 1.0
    resp = UnitHydrograph.from_array(kernel).convolve(impulse)
    np.testing.assert_allclose(resp[:n_ks], kernel, rtol=1e-12)
    np.testing.assert_allclose(resp[n_ks:], 0.0, atol=1e-15)
    # the same at a size where the long-series kernel runs, state carried across two calls
    n_ks, n, T = 48, 5000, 200
    kernel = synth.synth_uh_kernel(n, n_ks)
    depth = synth.synth_runoff_depth(n, 0, T)
    a = UnitHydrograph.from_array(kernel)
    first, second = a.convolve(depth[:120]), a.convolve(depth[120:])
    b = UnitHydrograph.from_array(kernel)
    whole = b.convolve(depth)
    np.testing.assert_allclose(np.vstack([first, second]), whole, rtol=1e-12, atol=1e-12 * np.abs(whole).max())
    np.testing.assert_allclose(a.state, b.state, rtol=1e-12, atol=1e-12 * np.abs(whole).max())
    # ... and convolve_incrementally (UnitHydrograph.py:64-75) against the ORACLE's definitional form, step by step with the state it
    # leaves behind (the comparison above is HIP against HIP): 30 steps of 300 basins, 12 taps, non-zero state carried in
    n_ks, n, T = 12, 300, 30
    kernel = synth.synth_uh_kernel(n, n_ks)
    depth = synth.synth_runoff_depth(n, 0, T)
    state0 = 1e-3 * np.random.default_rng(7).random((n_ks, n))
    uh, ref = UnitHydrograph.from_array(kernel), oracle.UnitHydrograph(kernel)
    uh.state[:] = state0
    ref.state[:] = state0
    for t in range(T):
        got, want = uh.convolve_incrementally(depth[t]), ref.convolve_incrementally(depth[t])
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12 * np.abs(want).max(), err_msg=f'step {t}')
        np.testing.assert_allclose(uh.state, ref.state, rtol=1e-12, atol=1e-12 * np.abs(ref.state).max(), err_msg=f'state after step {t}')


@pytest.mark.parametrize('n,T,nsub,factor', [(60_000, 256, 1, 1), (60_000, 256, 1, 4), (60_000, 96, 2, 2), (300_000, 384, 1, 8)])
def test_float32_output_fused_into_the_record_pass(monkeypatch, n, T, nsub, factor):
    """rr_rapid_route_f32_dev == float64 routing, then mean over `factor` rows, then the float32 cast
    (river_route/routers/TransformMuskingum.py:128-142), bit for bit; a factor that does not divide a batch is refused."""
    from river_route_amd._lib import RR_E_UNSUPPORTED, RRError
    set_env(monkeypatch, {})
    net = synth.synth_network(n, seed=17)
    indptr, indices = csc_from_down(net.down_index)
    dt = 900.0
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, dt)
    c4_dt = (c1 + c2) / (dt * nsub)
    ql = synth.synth_qlateral(n, 0, T, dt=dt * nsub)
    q0 = 2.0 * synth.u01(3, np.arange(n))
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(-c1[indices], c2, c3, c4_dt)
        d_ql, d_q = DeviceBuffer(ql.nbytes).upload(ql), DeviceBuffer(n * 8).upload(q0)
        d_64, d_32 = DeviceBuffer(T * n * 8), DeviceBuffer((T // factor) * n * 4)
        plan.rapid_route_dev(d_q, d_ql, T, d_64, T, T, nsub)
        q_a = d_q.download(np.float64, (n,))
        want = d_64.download(np.float64, (T, n)).reshape(T // factor, factor, n).mean(axis=1).astype(np.float32)
        d_q.upload(q0)
        plan.rapid_route_f32_dev(d_q, d_ql, T, d_32, T, nsub, factor)
        np.testing.assert_array_equal(d_32.download(np.float32, (T // factor, n)), want)
        np.testing.assert_array_equal(d_q.download(np.float64, (n,)), q_a)
        with pytest.raises(RRError) as e:
            plan.rapid_route_f32_dev(d_q, d_ql, T, d_32, T - T % 3, nsub, 3)
        assert e.value.code == RR_E_UNSUPPORTED
        for b in (d_ql, d_q, d_64, d_32):
            b.free()


@pytest.mark.parametrize('n,T,nsub,n_ks,env', [(60_000, 200, 1, 48, {}), (300_000, 150, 1, 48, {}), (60_000, 70, 3, 12, {}), (60_000, 40, 1, 60, {}),
                                               (60_000, 33, 1, 1, {}),
                                               (60_000, 400, 1, 48, {}), (60_000, 330, 1, 16, {}), (60_000, 130, 2, 60, {}),      # two pairs of record batches; a pair and a single one; sub-steps
                                               (60_000, 400, 1, 48, {'RR_UH_PAIRS': '0'}), (60_000, 130, 2, 60, {'RR_UH_PAIRS': '0'}),      # one batch per launch
                                               (60_007, 300, 1, 33, {}), (40_003, 100, 1, 5, {}),      # a last column tile of 7 and of 3 columns
                                               (1_000_000, 200, 1, 48, {})])      # BASELINE config 4's shape: the kernel bench.py times, k_rec_in_uh<false, 48, 2>
def test_unit_route_with_fused_convolution_vs_oracle(monkeypatch, n, T, nsub, n_ks, env):
    """rr_unit_route_uh_dev: UnitHydrograph.convolve + unit_route + the router's state bookkeeping
    (river_route/routers/UnitMuskingum.py:72-98) in one call, the convolved lateral never written as rows; two files, so
    the UH carry-over state (including T < n_ks leftovers) and the channel state both cross a file boundary."""
    set_env(monkeypatch, env)
    net = synth.synth_network(n, seed=23)
    indptr, indices = csc_from_down(net.down_index)
    hw_idx, inner_idx, A_in, A_hw = unit_split(indptr, indices, n)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0 / nsub)
    c1i, c2i, c3i = c1[inner_idx], c2[inner_idx], c3[inner_idx]
    args = (A_in.indptr, A_in.indices, -c1i[A_in.indices], A_in.indptr, A_in.indices, A_in.data,
            A_hw.indptr, A_hw.indices, A_hw.data, c1i, c2i, c3i, hw_idx, inner_idx)
    kern = synth.synth_uh_kernel(n, n_ks)
    uh = oracle.UnitHydrograph(kern)
    state_ref = 3.0 * synth.u01(7, np.arange(n))
    ni = inner_idx.size
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(-c1[indices], c2, c3, None)
        d_kern, d_state = DeviceBuffer(kern.nbytes).upload(kern), DeviceBuffer(kern.nbytes).upload(np.zeros_like(kern))
        d_depth, d_out, d_fin = DeviceBuffer(T * n * 8), DeviceBuffer(T * n * 8), DeviceBuffer(n * 8)
        d_qc, d_qf = DeviceBuffer(ni * 8), DeviceBuffer(ni * 8)
        state = state_ref.copy()
        for f, Tf in enumerate((T, max(2, n_ks // 2))):       # the second file is shorter than the kernel
            depth = synth.synth_runoff_depth(n, f * T, f * T + Tf)
            conv_ref = uh.convolve(depth)
            qc_ref, qf_ref, d_ref = state_ref[inner_idx].copy(), state_ref[inner_idx].copy(), np.zeros((Tf, n))
            oracle.unit_route(*args, qc_ref, qf_ref, conv_ref, d_ref, nsub)
            state_ref[hw_idx], state_ref[inner_idx] = conv_ref[-1][hw_idx], qf_ref
            d_depth.upload(depth)
            d_qc.upload(state[inner_idx].copy()); d_qf.upload(state[inner_idx].copy())
            try:
                plan.unit_route_uh_dev(d_qc, d_qf, d_fin, d_kern, d_state, n_ks, d_depth, Tf, nsub, discharge=d_out)
            except Exception as e:      # a call too short for the time-tiled kernel is refused: the two-call form then
                from river_route_amd._lib import RR_E_UNSUPPORTED
                assert getattr(e, 'code', None) == RR_E_UNSUPPORTED and Tf * nsub < 32
                break
            d = d_out.download(np.float64, (Tf, n))
            state = d_fin.download(np.float64, (n,))
            assert_close(d, d_ref, f'file {f} discharge')
            assert_close(state, state_ref, f'file {f} router state')
            assert_close(d_qc.download(np.float64, (ni,)), qc_ref, f'file {f} q_ch')
            assert_close(d_state.download(np.float64, kern.shape), uh.state, f'file {f} UH state')
        for b in (d_kern, d_state, d_depth, d_out, d_fin, d_qc, d_qf):
            b.free()


@pytest.mark.parametrize('wave,n,parts,T,nsub,chunk', [('1', 120_000, 4, 96, 1, 32), ('1', 60_000, 3, 40, 2, 16), ('0', 20_000, 5, 12, 2, 4),
                                                       ('1', 200_000, 8, 150, 1, 32)])
def test_unit_muskingum_on_a_cut_network_vs_oracle(monkeypatch, wave, n, parts, T, nsub, chunk):
    """UnitMuskingum with the network cut into parts (river_route/routers/_numba_kernels.py:88-171 on the undivided network
    is the oracle): ghosts of inner reaches and -- by moving one headwater across the cut -- a ghost of a headwater, carried
    state, each part convolving its own columns."""
    from river_route_amd.multi_gpu import HipUnitPartEngine, run_in_process, split_network
    set_env(monkeypatch, {'RR_WAVE': wave})
    net = synth.synth_network(n, seed=17)
    indptr, indices = csc_from_down(net.down_index)
    hw_idx, inner_idx, A_in, A_hw = unit_split(indptr, indices, n)
    dt = 3600.0 / nsub
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, dt)
    c1i, c2i, c3i = c1[inner_idx], c2[inner_idx], c3[inner_idx]
    args = (A_in.indptr, A_in.indices, -c1i[A_in.indices], A_in.indptr, A_in.indices, A_in.data,
            A_hw.indptr, A_hw.indices, A_hw.data, c1i, c2i, c3i, hw_idx, inner_idx)
    n_ks = 9
    kern = synth.synth_uh_kernel(n, n_ks)
    uh_state0 = 0.1 * synth.u01(3, np.arange(n_ks * n)).reshape(n_ks, n)
    depth = synth.synth_runoff_depth(n, 0, T)
    uh = oracle.UnitHydrograph(kern)
    uh.state = uh_state0.copy()
    conv_ref = uh.convolve(depth)
    q0 = 3.0 * synth.u01(7, np.arange(n))
    qc_ref, qf_ref, d_ref = 0.5 * q0[inner_idx], q0[inner_idx].copy(), np.zeros((T, n))
    oracle.unit_route(*args, qc_ref, qf_ref, conv_ref, d_ref, nsub)

    inner = np.zeros(n, dtype=bool)
    inner[inner_idx] = True
    part_of, _ = partition_forest(indptr, indices, parts)
    part_of = part_of.copy()
    last = np.flatnonzero(~inner & (net.down_index >= 0) & (part_of == parts - 1))
    part_of[last[0]] = 0                                   # a cut directly below a headwater
    specs = [split_network(net.down_index, part_of, p, parts, inner_global=inner) for p in range(parts)]
    assert sum(s.n_dummy for s in specs) > 0 and sum(s.n_ghost - s.n_dummy for s in specs) > 0
    engines = [HipUnitPartEngine(s, c1, c2, c3, inner, 0.5 * q0[inner_idx], q0[inner_idx], kern[:, s.real_global],
                                 uh_state0[:, s.real_global], depth[:, s.real_global], T, nsub, 0) for s in specs]
    run_in_process(engines, specs, T, nsub, chunk)
    if wave == '1':
        assert all(e.plan.profile()['ticks_per_launch'] >= 16 for e in engines)      # the time-tiled kernel ran
    qc, qf = np.full(inner_idx.size, np.nan), np.full(inner_idx.size, np.nan)
    for s, e in zip(specs, engines):
        d = e.discharge.cpu().numpy()[:, s.n_lead:]
        assert_close(d, d_ref[:, s.real_global], f'part {s.part} discharge')
        rank, a, b = e.final_state()
        qc[rank], qf[rank] = a, b
        np.testing.assert_allclose(e.uh_state.cpu().numpy(), uh.state[:, s.real_global], rtol=0, atol=1e-12 * np.abs(uh.state).max())
    assert_close(qc, qc_ref, 'q_ch')
    assert_close(qf, qf_ref, 'q_full')


@pytest.mark.parametrize('n,T', [(1_000_000, 35_040), (4_000_000, 8_760), (250_000, 35_040), (100_000, 20_000)])
def test_full_year_at_1m_time_tiled_equals_streaming(monkeypatch, n, T):
    """BASELINE config 3 at full length (1M reaches x 35,040 steps, the bench's cyclic forcing and sink): the record ring
    goes round eight times; 4M reaches, where the ring is as large as the card allows and the tasks are 16 ticks; and two smaller
    networks, whose long calls get tasks of 256 and 128 ticks.  The time-tiled path (k_tile + record passes) and the streaming path (k_tick, no ring) evaluate
    the same expression in the same order, so the final state and the last 96 discharge rows must agree bit for bit; the
    streaming kernel is the one compared with the oracle row by row elsewhere."""
    import torch
    rows = 96
    net = synth.synth_network(n)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0)
    dev = torch.device('cuda:0')
    ql = torch.from_numpy(synth.synth_qlateral(n, 0, rows)).to(dev)
    got = {}
    for wave in ('1', '0'):
        set_env(monkeypatch, {'RR_WAVE': wave})
        with Plan(indptr, indices) as plan:
            plan.set_coeffs(-c1[indices], c2, c3, (c1 + c2) / 900.0)
            q = torch.zeros(n, dtype=torch.float64, device=dev)
            out = torch.zeros((rows, n), dtype=torch.float64, device=dev)
            plan.rapid_route_dev(q, ql, rows, out, rows, T, 1, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert (plan.profile()['ticks_per_launch'] >= 16) == (wave == '1')
            got[wave] = (q.cpu().numpy(), out.cpu().numpy())
    assert np.isfinite(got['1'][0]).all() and got['1'][0].max() > 0
    np.testing.assert_array_equal(got['1'][0], got['0'][0])
    np.testing.assert_array_equal(got['1'][1], got['0'][1])


@pytest.mark.parametrize('wave', ['1', '0'])
def test_export_reach_with_a_downstream_reach_in_the_plan(monkeypatch, wave):
    """rr_plan_set_boundary accepts any reach as an export reach.  One that still has its downstream reach in the plan (not
    an outlet of a part, as the partitioner makes them) may be mirrored by a tile ghost, whose slot the export index would
    need: such a plan keeps to the streaming kernel.  Either way the export series is the reach's unclamped discharge after
    every sub-step and the routed rows match the oracle."""
    set_env(monkeypatch, {'RR_WAVE': wave})
    n, T, nsub = 40_000, 48, 2
    net = synth.synth_network(n, seed=5)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 450.0)
    c4_dt = (c1 + c2) / 450.0
    ql = synth.synth_qlateral(n, 0, T)
    upstream_count = np.bincount(net.down_index[net.down_index >= 0], minlength=n)
    inner_with_down = np.flatnonzero((upstream_count > 0) & (net.down_index >= 0))
    exports = inner_with_down[[5, len(inner_with_down) // 2, -3]]
    q_ref = np.zeros(n)
    series_ref = np.zeros((T * nsub, exports.size))
    for t in range(T):      # the oracle row by row, sub-step by sub-step, for the unclamped values
        for s in range(nsub):
            tmp = np.zeros((1, n))
            oracle.rapid_route(indptr, indices, -c1[indices], c2, c3, c4_dt, q_ref, ql[t:t + 1], tmp, 1)
            series_ref[t * nsub + s] = q_ref[exports]
    q_chk, d_chk = np.zeros(n), np.zeros((T, n))
    oracle.rapid_route(indptr, indices, -c1[indices], c2, c3, c4_dt, q_chk, ql, d_chk, nsub)
    assert_close(q_chk, q_ref, 'oracle stepwise == oracle joint')
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(-c1[indices], c2, c3, c4_dt)
        plan.set_boundary([], exports)
        d_q, d_ql, d_out = DeviceBuffer(n * 8).upload(np.zeros(n)), DeviceBuffer(ql.nbytes).upload(ql), DeviceBuffer(T * n * 8)
        d_exp = DeviceBuffer(T * nsub * exports.size * 8)
        plan.stream_begin(d_q, d_ql, T, d_out, T, T, nsub, None, d_exp)
        plan.stream_advance(T, T * nsub)
        plan.stream_end(d_q)
        assert_close(d_out.download(np.float64, (T, n)), d_chk, 'discharge')
        assert_close(d_q.download(np.float64, (n,)), q_chk, 'state')
        assert_close(d_exp.download(np.float64, (T * nsub, exports.size)), series_ref, 'export series')
        for b in (d_q, d_ql, d_out, d_exp):
            b.free()


@pytest.mark.parametrize('p_chain,n_outlets,p_third', [(0.999, 3, 0.02), (0.97, 40, 0.3)])
def test_chain_grown_forests_vs_oracle(monkeypatch, p_chain, n_outlets, p_third):
    """SURVEY section 8(d)'s generator (synth_network_chain): long in-degree-1 runs, several outlets, in-degree 3 -- the shapes
    real networks have and the Remy tree does not.  200k reaches more than 10k deep (and a shallow forest with many
    three-way confluences) through the time-tiled kernel the engine picks by itself, against the oracle."""
    set_env(monkeypatch, {})
    n, T = 200_000, 96
    net = synth.synth_network_chain(n, seed=77, p_chain=p_chain, n_outlets=n_outlets, p_third=p_third)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0)
    lhs, c4_dt = -c1[indices], (c1 + c2) / 900.0
    ql = synth.synth_qlateral(n, 0, T)
    q0 = 2.0 * synth.u01(3, np.arange(n))
    q_ref, d_ref = q0.copy(), np.zeros((T, n))
    oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, ql, d_ref, 1)
    with Plan(indptr, indices) as plan:
        if p_chain > 0.99:
            assert plan.depth > 10_000
        assert plan.n_outlets == n_outlets
        plan.set_coeffs(lhs, c2, c3, c4_dt)
        d_q, d_ql, d_out = DeviceBuffer(n * 8).upload(q0), DeviceBuffer(ql.nbytes).upload(ql), DeviceBuffer(T * n * 8)
        plan.rapid_route_dev(d_q, d_ql, T, d_out, T, T, 1)
        prof = plan.profile()
        assert prof['ticks_per_launch'] > 1, 'the time-tiled kernel was expected'
        assert_close(d_out.download(np.float64, (T, n)), d_ref, 'discharge')
        assert_close(d_q.download(np.float64, (n,)), q_ref, 'q_t')
        for b in (d_q, d_ql, d_out):
            b.free()


@pytest.mark.parametrize('n,T,nsub,factor', [(100_000, 256, 1, 1), (100_000, 256, 1, 4), (50_001, 128, 2, 2)])
def test_float32_lateral_rows_equal_their_float64_copy(monkeypatch, n, T, nsub, factor):
    """rr_rapid_route_f32in_dev: float32 lateral rows converted inside the in-pass.  Both output forms, against
    rr_rapid_route_dev / rr_rapid_route_f32_dev on the float64 copy of the same values: bit for bit."""
    set_env(monkeypatch, {})
    net = synth.synth_network(n, seed=41)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0 / nsub)
    ql32 = synth.synth_qlateral(n, 0, T).astype(np.float32)
    ql64 = ql32.astype(np.float64)
    q0 = 2.0 * synth.u01(3, np.arange(n))
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(-c1[indices], c2, c3, (c1 + c2) / 900.0)
        d_ql32, d_ql64 = DeviceBuffer(ql32.nbytes).upload(ql32), DeviceBuffer(ql64.nbytes).upload(ql64)
        d_q, d_a, d_b = DeviceBuffer(n * 8), DeviceBuffer(T * n * 8), DeviceBuffer(T * n * 8)
        d_q.upload(q0); plan.rapid_route_dev(d_q, d_ql64, T, d_a, T, T, nsub)
        qa = d_q.download(np.float64, (n,))
        d_q.upload(q0); plan.rapid_route_f32in_dev(d_q, d_ql32, T, T, nsub, discharge=d_b, out_rows=T)
        np.testing.assert_array_equal(d_b.download(np.float64, (T, n)), d_a.download(np.float64, (T, n)))
        np.testing.assert_array_equal(d_q.download(np.float64, (n,)), qa)
        d_q.upload(q0); plan.rapid_route_f32_dev(d_q, d_ql64, T, d_a, T, nsub, factor)
        d_q.upload(q0); plan.rapid_route_f32in_dev(d_q, d_ql32, T, T, nsub, discharge32=d_b, factor=factor)
        np.testing.assert_array_equal(d_b.download(np.float32, (T // factor, n)), d_a.download(np.float32, (T // factor, n)))
        # and against the oracle on the float64 copy
        q_ref, d_ref = q0.copy(), np.zeros((T, n))
        oracle.rapid_route(indptr, indices, -c1[indices], c2, c3, (c1 + c2) / 900.0, q_ref, ql64, d_ref, nsub)
        d_q.upload(q0); plan.rapid_route_f32in_dev(d_q, d_ql32, T, T, nsub, discharge=d_b, out_rows=T)
        assert_close(d_b.download(np.float64, (T, n)), d_ref, 'discharge')
        for b in (d_ql32, d_ql64, d_q, d_a, d_b):
            b.free()


def _wide_network(n, seed=5, fan=8):
    """Reach i flows into one of the next `fan` reaches: in-degrees are Poisson-like, several per cent of the reaches have four
    or more upstream reaches (no synthetic family above has more than three), the last one is the outlet."""
    rng = np.random.default_rng(seed)
    down = np.arange(n) + 1 + rng.integers(0, fan, n)
    down[down >= n] = n - 1
    down[n - 1] = -1
    return down.astype(np.int64)


@pytest.mark.parametrize('env', [{}, {'RR_TILE_LEAN': '0'}])
def test_confluences_of_four_and_more_vs_oracle(monkeypatch, env):
    """The short tick reads three upstream values; a tile that holds a reach with more goes to the companion launch of the
    general kernel in the same step (TileArgs::tile_filter).  Also: the general kernel alone (RR_TILE_LEAN=0) on the same network."""
    set_env(monkeypatch, env)
    n, T = 150_000, 130
    down = _wide_network(n)
    indeg = np.bincount(down[down >= 0], minlength=n)
    assert indeg.max() >= 5 and (indeg >= 4).mean() > 0.005
    indptr, indices = csc_from_down(down)
    rng = np.random.default_rng(1)
    k, x = rng.uniform(900.0, 7200.0, n), rng.uniform(0.05, 0.45, n)
    c1, c2, c3 = oracle.muskingum_coefficients(k, x, 900.0)
    lhs, c4_dt = -c1[indices], (c1 + c2) / 900.0
    ql = synth.synth_qlateral(n, 0, T)
    q0 = 2.0 * synth.u01(3, np.arange(n))
    q_ref, d_ref = q0.copy(), np.zeros((T, n))
    oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, ql, d_ref, 1)
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(lhs, c2, c3, c4_dt)
        d_q, d_ql, d_out = DeviceBuffer(n * 8).upload(q0), DeviceBuffer(ql.nbytes).upload(ql), DeviceBuffer(T * n * 8)
        for rep in range(2):      # the second call continues from the first one's state
            plan.rapid_route_dev(d_q, d_ql, T, d_out, T, T, 1)
            assert plan.profile()['ticks_per_launch'] > 1
            assert_close(d_out.download(np.float64, (T, n)), d_ref, f'discharge, call {rep}')
            assert_close(d_q.download(np.float64, (n,)), q_ref, f'q_t, call {rep}')
            oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, ql, d_ref, 1)
        for b in (d_q, d_ql, d_out):
            b.free()


def test_record_ring_goes_round_three_times_vs_oracle(monkeypatch):
    """6,720 rows over a 200k-reach network: the record ring (depth + levels x K tick-rows, 2,048 here) is reused three times, the
    forcing is a 96-row cyclic array and the discharge goes to a 128-row cyclic sink (one out-pass batch, as in bench.py) --
    against the oracle carried through the same 70 passes over the forcing: the rows left in the sink are the oracle's last
    128, the state its final state.  (At 1M reaches x 35,040 rows the same wrap-around is compared with the streaming kernel
    bit for bit, test_full_year_at_1m_time_tiled_equals_streaming; this is the oracle's word on it at a size it finishes in seconds.)"""
    set_env(monkeypatch, {})
    n, rows, passes, sink = 200_000, 96, 70, 128
    T = rows * passes
    net = synth.synth_network(n, seed=9)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0)
    lhs, c4_dt = -c1[indices], (c1 + c2) / 900.0
    ql = synth.synth_qlateral(n, 0, rows)
    q_ref, d_now, d_before = np.zeros(n), np.zeros((rows, n)), np.zeros((rows, n))
    for _ in range(passes):
        d_before, d_now = d_now, d_before
        oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, ql, d_now, 1)
    last = np.concatenate([d_before, d_now])[-sink:]                      # rows T - 128 ... T - 1
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(lhs, c2, c3, c4_dt)
        from river_route_amd.engine import MODE_RAPID
        sch = plan.reserve(MODE_RAPID, T, 1)
        assert sch['tiled'] and sch['ring_chunks'] * 16 * 3 < T, sch      # three revolutions
        d_q, d_ql, d_out = DeviceBuffer(n * 8).upload(np.zeros(n)), DeviceBuffer(ql.nbytes).upload(ql), DeviceBuffer(sink * n * 8)
        plan.rapid_route_dev(d_q, d_ql, rows, d_out, sink, T, 1)
        got = d_out.download(np.float64, (sink, n))[(np.arange(T - sink, T) % sink)]      # row t sits at t % 128
        assert_close(got, last, 'last 128 rows')
        assert_close(d_q.download(np.float64, (n,)), q_ref, 'final state')
        for b in (d_q, d_ql, d_out):
            b.free()


def _sub_basins(down, lo, hi, want):
    """`want` disjoint sub-basins (a reach and everything upstream of it) of lo..hi reaches, spread over the index range:
    (columns ascending, down index within them)."""
    n = down.size
    dl, size = down.tolist(), [1] * n
    for i in range(n):                           # upstream reaches come first (tools.py:103-104)
        if dl[i] >= 0:
            size[dl[i]] += size[i]
    size = np.asarray(size)
    below = np.where(down >= 0, size[np.maximum(down, 0)], hi + 1)
    roots = np.flatnonzero((size >= lo) & (size <= hi) & (below > hi))      # the largest such basins: none inside another
    assert roots.size >= want
    roots = roots[np.linspace(0, roots.size - 1, want).astype(np.int64)]
    is_root = np.zeros(n, dtype=bool)
    is_root[roots] = True
    jump = np.where(is_root | (down < 0), np.arange(n), down)
    while True:                                  # pointer doubling: every reach ends at the first chosen root below it, or at its outlet
        nxt = jump[jump]
        if np.array_equal(nxt, jump):
            break
        jump = nxt
    cols = np.flatnonzero(is_root[jump])
    assert cols.size == size[roots].sum()
    sub_down = np.where(is_root[cols], -1, np.searchsorted(cols, down[cols]))
    return cols, sub_down


def test_full_year_at_1m_sub_basins_vs_oracle(monkeypatch):
    """BASELINE config 3 at full length against the ORACLE: discharge of a reach depends on its sub-basin only, so the oracle
    routes four sub-basins (3k-6k reaches each) of the 1M-reach network on their own through all 35,040 steps, and the engine's
    rows for those columns -- from two calls of 17,520 rows over the whole network (140 GB of discharge rows each, the record
    ring of 6,000 tick-rows gone round almost three times per call, state carried over, the schedule of the headline: 128 ticks per task) -- must be theirs row
    by row.  Complements test_full_year_at_1m_time_tiled_equals_streaming, which covers every column but against k_tick."""
    import torch
    set_env(monkeypatch, {})
    n, T, rows, calls = 1_000_000, 35_040, 120, 2      # 146 passes over the forcing per call
    net = synth.synth_network(n)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0)
    lhs, c4_dt = -c1[indices], (c1 + c2) / 900.0
    cols, sub_down = _sub_basins(net.down_index, 3_000, 6_000, 4)
    s_indptr, s_indices = csc_from_down(sub_down)
    dev = torch.device('cuda:0')
    ql = synth.synth_qlateral_torch(n, 0, rows, dev)
    cols_t = torch.from_numpy(cols).to(dev)
    ql_sub = ql[:, cols_t].cpu().numpy()
    s_c1, s_c2, s_c3, s_c4 = c1[cols], np.ascontiguousarray(c2[cols]), np.ascontiguousarray(c3[cols]), np.ascontiguousarray(c4_dt[cols])
    s_lhs = -s_c1[s_indices]
    q_ref, d_ref = np.zeros(cols.size), np.zeros((rows, cols.size))
    Tc = T // calls
    assert Tc % rows == 0
    worst = 0.0
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(lhs, c2, c3, c4_dt)
        from river_route_amd.engine import MODE_RAPID
        sch = plan.reserve(MODE_RAPID, Tc, 1)
        q = torch.zeros(n, dtype=torch.float64, device=dev)
        out = torch.empty((Tc, n), dtype=torch.float64, device=dev)
        for call in range(calls):
            out.fill_(-1.0)
            plan.rapid_route_dev(q, ql, rows, out, Tc, Tc, 1, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert plan.profile()['ticks_per_launch'] == 128 and sch['ring_chunks'] * 16 * 2 < Tc, sch
            kept = out[:, cols_t]                                                 # (17,520, ~18k)
            for r0 in range(0, Tc, rows):
                oracle.rapid_route(s_indptr, s_indices, s_lhs, s_c2, s_c3, s_c4, q_ref, ql_sub, d_ref, 1)
                got = kept[r0:r0 + rows].cpu().numpy()
                scale = np.abs(d_ref).max()
                err = np.abs(got - d_ref).max() / scale
                worst = max(worst, err)
                assert err <= 1e-10, f'call {call}, rows {r0}..{r0 + rows}: {err:.3e} of the largest discharge'
            del kept
        assert_close(q[cols_t].cpu().numpy(), q_ref, 'final state of the sub-basins')
    print(f'sub-basins: {cols.size} reaches x {T} steps against the oracle, worst difference {worst:.2e} of the largest discharge')


def test_constant_forcing_settles_at_the_basin_sums(monkeypatch):
    """A size-independent property at full size, no oracle involved: under a forcing that does not change, the update
    q+ = c3 q + c4 ql/dt + c2 A q + c1 A q+ (_numba_kernels.py:68-78) has the fixed point q = A q + ql/dt because
    1 - c3 = c1 + c2 = c4 (Muskingum.py:174-179): every reach ends at the lateral inflow accumulated over its basin.  One year
    over 1M reaches (one forcing row read 35,040 times, the ring reused eight times) must have arrived there at EVERY reach,
    outlet included."""
    import torch
    set_env(monkeypatch, {})
    n, T, sink = 1_000_000, 35_040, 128
    net = synth.synth_network(n)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0)
    row = synth.synth_qlateral(n, 0, 1)
    acc, dl = (row[0] / 900.0).tolist(), net.down_index.tolist()
    for i in range(n):
        if dl[i] >= 0:
            acc[dl[i]] += acc[i]
    want = np.asarray(acc)
    dev = torch.device('cuda:0')
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(-c1[indices], c2, c3, (c1 + c2) / 900.0)
        q = torch.zeros(n, dtype=torch.float64, device=dev)
        out = torch.zeros((sink, n), dtype=torch.float64, device=dev)
        plan.rapid_route_dev(q, torch.from_numpy(row).to(dev), 1, out, sink, T, 1, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert plan.profile()['ticks_per_launch'] >= 64
        np.testing.assert_allclose(q.cpu().numpy(), want, rtol=1e-9, err_msg='final state')
        np.testing.assert_allclose(out.cpu().numpy(), np.broadcast_to(want, (sink, n)), rtol=1e-9, err_msg='last 128 rows')


@pytest.mark.parametrize('n,T,nsub,n_ks,factor', [(60_007, 300, 1, 48, 1), (60_000, 130, 2, 12, 0), (100_000, 384, 1, 33, 4)])
def test_float32_depth_rows_equal_their_float64_copy(monkeypatch, n, T, nsub, n_ks, factor):
    """rr_unit_route_uh_f32in_dev: runoff depths as float32 rows (4 bytes read per value) give, bit for bit, what the same values
    give as float64 rows through rr_unit_route_uh_dev -- discharge (float64 rows, or float32 rows averaged by `factor`), the
    router's state, q_ch and the UH carry-over state -- over two files, the second shorter than the kernel."""
    set_env(monkeypatch, {})
    net = synth.synth_network(n, seed=29)
    indptr, indices = csc_from_down(net.down_index)
    hw_idx, inner_idx, A_in, A_hw = unit_split(indptr, indices, n)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0 / nsub)
    kern = synth.synth_uh_kernel(n, n_ks)
    ni = inner_idx.size
    seed = 3.0 * synth.u01(7, np.arange(n))
    results = {}
    for kind in ('f4', 'f8'):
        with Plan(indptr, indices) as plan:
            plan.set_coeffs(-c1[indices], c2, c3, None)
            d_kern, d_state = DeviceBuffer(kern.nbytes).upload(kern), DeviceBuffer(kern.nbytes).upload(np.zeros_like(kern))
            d_fin, d_qc, d_qf = DeviceBuffer(n * 8), DeviceBuffer(ni * 8), DeviceBuffer(ni * 8)
            state, got = seed.copy(), []
            for f, Tf in enumerate((T, max(2, n_ks // 2))):
                if factor and Tf % factor:
                    Tf -= Tf % factor
                depth32 = synth.synth_runoff_depth(n, f * T, f * T + Tf).astype(np.float32)
                d_qc.upload(state[inner_idx].copy()); d_qf.upload(state[inner_idx].copy())
                rows_out = Tf // factor if factor else Tf
                d_out = DeviceBuffer(rows_out * n * (4 if factor else 8))
                out = dict(discharge32=d_out, factor=factor) if factor else dict(discharge=d_out)
                try:
                    if kind == 'f4':
                        d_depth = DeviceBuffer(depth32.nbytes).upload(depth32)
                        plan.unit_route_uh_f32in_dev(d_qc, d_qf, d_fin, d_kern, d_state, n_ks, d_depth, Tf, nsub, **out)
                    else:
                        depth64 = depth32.astype(np.float64)
                        d_depth = DeviceBuffer(depth64.nbytes).upload(depth64)
                        plan.unit_route_uh_dev(d_qc, d_qf, d_fin, d_kern, d_state, n_ks, d_depth, Tf, nsub, **out)
                except Exception as e:      # a call too short for the time-tiled kernel is refused by both forms alike
                    from river_route_amd._lib import RR_E_UNSUPPORTED
                    assert getattr(e, 'code', None) == RR_E_UNSUPPORTED
                    got.append(None)
                    break
                state = d_fin.download(np.float64, (n,))
                got.append((d_out.download(np.float32 if factor else np.float64, (rows_out, n)), state.copy(),
                            d_qc.download(np.float64, (ni,)), d_state.download(np.float64, kern.shape)))
                d_depth.free(); d_out.free()
            for b in (d_kern, d_state, d_fin, d_qc, d_qf):
                b.free()
        results[kind] = got
    assert len(results['f4']) == len(results['f8']) and results['f4'][0] is not None
    for a, b in zip(results['f4'], results['f8']):
        assert (a is None) == (b is None)
        if a is not None:
            for x, y in zip(a, b):
                np.testing.assert_array_equal(x, y)


def test_unit_full_year_as_ten_calls_sub_basins_vs_oracle(monkeypatch):
    """BASELINE config 4 at full length, as bench.py's `year` field times it: ten consecutive calls of 3,504 rows of runoff depths at
    1M reaches with the 48-step kernel (the reference's loop over runoff files, UnitMuskingum.py:72-98), the convolution's tail
    (UnitHydrograph.py:99-105) and the channel state (q_ch, q_full) carried from call to call -- against the ORACLE on four
    sub-basins (a reach's discharge depends on its sub-basin only; the convolution is column-wise) through all 35,040 steps: the
    oracle convolves and routes the sub-basins' columns call by call with its own carried state."""
    import torch
    set_env(monkeypatch, {})
    n, T, calls, n_ks, dt = 1_000_000, 3_504, 10, 48, 900.0
    net = synth.synth_network(n)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, dt)
    cols, sub_down = _sub_basins(net.down_index, 2_000, 5_000, 4)
    s_indptr, s_indices = csc_from_down(sub_down)
    ns = cols.size
    hw_idx, inner_idx, A_in, A_hw = unit_split(s_indptr, s_indices, ns)
    s_c1, s_c2, s_c3 = c1[cols], c2[cols], c3[cols]
    c1i, c2i, c3i = s_c1[inner_idx], s_c2[inner_idx], s_c3[inner_idx]
    kern = synth.synth_uh_kernel(n, n_ks, tr=dt)
    dev = torch.device('cuda:0')
    cols_t = torch.from_numpy(cols).to(dev)
    ref_uh = oracle.UnitHydrograph(np.ascontiguousarray(kern[:, cols]))
    qc_ref, qf_ref = np.zeros(inner_idx.size), np.zeros(inner_idx.size)
    d_kern = torch.from_numpy(kern).to(dev)
    d_state = torch.zeros_like(d_kern)
    worst = 0.0
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(-c1[indices], c2, c3, None)
        n_inner = plan.n_inner
        q_ch = torch.zeros(n_inner, dtype=torch.float64, device=dev)
        q_full = torch.zeros(n_inner, dtype=torch.float64, device=dev)
        q_final = torch.zeros(n, dtype=torch.float64, device=dev)
        out = torch.empty((T, n), dtype=torch.float64, device=dev)
        g = torch.Generator(device=dev)
        for call in range(calls):
            g.manual_seed(900 + call)      # every file its own depths
            depth = torch.rand((T, n), dtype=torch.float64, device=dev, generator=g) * 1e-3
            out.fill_(-1.0)
            plan.unit_route_uh_dev(q_ch, q_full, q_final, d_kern, d_state, n_ks, depth, T, 1, discharge=out, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert plan.last_kernel() == 'tile'
            got = out[:, cols_t].cpu().numpy()
            conv = ref_uh.convolve(depth[:, cols_t].cpu().numpy())
            d_ref = np.zeros((T, ns))
            oracle.unit_route(A_in.indptr, A_in.indices, -c1i[A_in.indices], A_in.indptr, A_in.indices, A_in.data, A_hw.indptr, A_hw.indices, A_hw.data,
                              c1i, c2i, c3i, hw_idx, inner_idx, qc_ref, qf_ref, conv, d_ref, 1)
            # the router re-seeds q_ch from the state it keeps (q_full on inner reaches) at every file: UnitMuskingum.py:78-79
            qc_ref[:] = qf_ref
            q_ch.copy_(q_full)
            scale = float(np.abs(d_ref).max())
            err = float(np.abs(got - d_ref).max()) / scale
            worst = max(worst, err)
            assert err <= 1e-10, f'call {call}: {err:.3e} of the largest discharge'
            del depth
        st = d_state[:, cols_t].cpu().numpy()
        np.testing.assert_allclose(st, ref_uh.state, rtol=1e-10, atol=1e-10 * float(np.abs(ref_uh.state).max()), err_msg='carried convolution tail')
    del out, d_kern, d_state
    torch.cuda.empty_cache()
    print(f'UnitMuskingum sub-basins: {ns} reaches x {calls * T} steps in {calls} calls against the oracle, worst difference {worst:.2e} of the largest discharge')

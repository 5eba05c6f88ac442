"""The direct row path (rr::DirectPlan, k_direct): host-side invariants of the column-range layout (no GPU: plans
are host-only) and, on the GPU box, the path against the oracle -- network sizes up to 1M reaches, tasks of 64 to 512 rows, cyclic
forcing arrays, the skeleton's record ring gone round several times, consecutive calls -- and against the record path bit for bit."""
import numpy as np
import pytest

from conftest import assert_close
from oracle import oracle
from river_route_amd import synth
from river_route_amd._lib import RR_DEVICE_NONE
from river_route_amd.engine import DeviceBuffer, Plan

HOLE, GHOST, MASK = 1 << 30, 1 << 28, (1 << 27) - 1
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


def check_direct_layout(down):
    n = down.shape[0]
    indptr, indices = csc_from_down(down)
    with Plan(indptr, indices, device=RR_DEVICE_NONE) as plan:
        info = plan.direct_info()
        assert info['ok'], info['why']
        L = plan.direct_layout()
        perm_h, lag_h, _ = plan.layout()
    lag_of = np.empty(n, np.int64)
    lag_of[perm_h] = lag_h
    c0, nc, lo, span, dword, up3, xinfo = (L[k].astype(np.int64) for k in ('tile_c0', 'tile_nc', 'tile_lag_lo', 'tile_span', 'delay', 'up3', 'xinfo'))
    delay, sender = (dword & 0xFF) | (dword & HOLE), (dword >> 8) & 0x7F      # the delay word also carries 1 + the column's number among its tile's senders
    # the tiles are consecutive column ranges that cover every column once
    assert c0[0] == 0 and np.array_equal(c0[1:], (c0 + nc)[:-1]) and c0[-1] + nc[-1] == n and nc.min() >= 1 and nc.max() <= 256
    assert span.max() + 3 == info['window_rows'] <= 72      # the LDS window: span + 1 rows in flight, one arriving, one leaving
    tile_of = np.repeat(np.arange(c0.size), nc)
    hole = (delay & HOLE) != 0
    assert hole.sum() == info['holes']
    small = ~hole
    # a lane's delay is its lag above the tile's smallest
    assert np.array_equal(delay[small], lag_of[small] - lo[tile_of[small]])
    assert np.all(delay[small] >= 0) and np.all(delay[small] <= span[tile_of[small]])
    for t in np.unique(tile_of[small]):
        d = delay[small & (tile_of == t)]
        assert d.min() == 0 and d.max() == span[t]
    # the upstream lanes of a lane: exactly the reaches that flow into it, in its tile, one tick ahead, headwaters first
    n_up = np.bincount(down[down >= 0], minlength=n)
    lanes = np.stack([(up3 >> (10 * k)) & 0x3FF for k in range(3)], axis=1)
    cnt = (lanes != 0x3FF).sum(axis=1)
    assert np.array_equal(cnt[small], n_up[small]) and np.all(cnt[hole] == 0)
    for k in range(3):
        sel = small & (lanes[:, k] != 0x3FF)
        u = c0[tile_of[sel]] + lanes[sel, k]
        assert np.array_equal(down[u], np.flatnonzero(sel)) and np.array_equal(tile_of[u], tile_of[sel])
        assert np.array_equal(delay[u], delay[sel] - 1)
        if k:      # filled from slot 0; a headwater never follows a reach with tributaries
            prev = c0[tile_of[sel]] + lanes[sel, k - 1]
            assert np.all(lanes[sel, k - 1] != 0x3FF) and np.all((n_up[prev] == 0) | (n_up[u] > 0))
    # a hole is a reach with a large or tall subtree; its downstream reach is one too; a small reach below ... is an outlet
    assert np.all(hole[down[hole & (down >= 0)]])
    outlet = small & (down >= 0) & hole[np.maximum(down, 0)]
    assert outlet.sum() == info['outlets'] and np.all(xinfo[outlet] >= 0) and np.all(xinfo[small & ~outlet] == -1)
    assert np.unique(np.concatenate([xinfo[hole], xinfo[outlet]])).size == hole.sum() + outlet.sum()
    sends = hole | outlet      # numbered 1, 2, ... in column order inside each tile, at most 64
    assert np.all(sender[~sends] == 0) and sender.max(initial=0) <= 64
    if sends.any():
        first = np.r_[True, tile_of[np.flatnonzero(sends)][1:] != tile_of[np.flatnonzero(sends)][:-1]]
        assert np.all(sender[sends][first] == 1) and np.all(np.diff(sender[sends])[~first[1:]] == 1)
    # the skeleton's positions: its reaches, one ghost per outlet that feeds them, and the ghosts between its own pieces
    assert info['skeleton_positions'] >= hole.sum() + outlet.sum() and np.concatenate([xinfo[hole], xinfo[outlet]]).max(initial=-1) < info['skeleton_positions']
    return info, dict(tile_of=tile_of, hole=hole)


@pytest.mark.parametrize('n,seed', [(9, 1), (300, 1), (5000, 2), (60000, 21), (250000, 13)])
def test_postorder_networks_tile_into_column_ranges(n, seed):
    info, _ = check_direct_layout(synth.synth_network(n, seed=seed, order='postorder').down_index)
    if n >= 60000:
        assert info['holes'] <= 0.06 * n and info['tiles'] <= 1.35 * n / 256
        assert info['skeleton_levels'] <= 26


def test_postorder_forest_with_three_way_confluences():
    net = synth.synth_network_chain(40_000, p_chain=0.3, n_outlets=7, p_third=0.05, seed=5, order='postorder')
    check_direct_layout(net.down_index)


@pytest.mark.parametrize('n,chainy', [(1, False), (9, False), (5000, False), (120_000, False), (40_000, True)])
def test_tools_postorder_sorts_any_table_into_column_range_order(n, chainy):
    """tools.postorder on a network table whose rows come in ANY order (not even topologically sorted): the re-sorted table is
    accepted by adjacency_matrix (river_route/tools.py:103-104), every sub-basin is a run of consecutive rows, and the plan built on
    it takes the direct row path."""
    from river_route_amd import tools
    net = (synth.synth_network_chain(n, p_chain=0.4, n_outlets=6, p_third=0.05, seed=8) if chainy else synth.synth_network(n, seed=6))
    shuffle = np.random.default_rng(5).permutation(n)
    rid, did = net.river_ids[shuffle], net.downstream_ids[shuffle]
    order = tools.postorder(rid, did)
    assert np.array_equal(np.sort(order), np.arange(n))
    rid, did = rid[order], did[order]
    A = tools.adjacency_matrix(rid, did)      # raises unless upstream comes before downstream
    down = np.full(n, -1, dtype=np.int64)
    coo = A.tocoo()
    down[coo.col] = coo.row
    sub = np.ones(n, dtype=np.int64)
    first = np.arange(n)
    for c in range(n):      # upstream first
        if down[c] >= 0:
            sub[down[c]] += sub[c]
            first[down[c]] = min(first[down[c]], first[c])
    assert np.array_equal(first, np.arange(n) - sub + 1), 'every sub-basin is the run of rows that ends at its outlet'
    if n >= 5000:
        info, _ = check_direct_layout(down)
        assert info['ok']


def test_tools_postorder_puts_a_main_stem_side_by_side():
    """tools.postorder visits a reach's small tributaries first and its main stem last, so the reaches with large sub-basins -- the direct
    row path's holes, patched into the output rows by 8-byte stores -- lie in runs: far fewer 64-byte lines of a row hold one than there
    are holes (rr_plan.cpp: postorder; 1M reaches: 7,375 lines for 50,021 holes, 26,192 with the largest tributary first)."""
    from river_route_amd import tools
    n = 120_000
    net = synth.synth_network(n, seed=6, order='random')
    order = tools.postorder(net.river_ids, net.downstream_ids)
    rid, did = net.river_ids[order], net.downstream_ids[order]
    A = tools.adjacency_matrix(rid, did)
    down = np.full(n, -1, dtype=np.int64)
    coo = A.tocoo()
    down[coo.col] = coo.row
    info, L = check_direct_layout(down)
    holes = np.nonzero(L['hole'])[0]
    lines = np.unique(holes >> 3).size
    runs = 1 + np.count_nonzero(np.diff(holes) > 1)
    assert info['holes'] == holes.size > 4000
    assert lines < 0.2 * holes.size and runs < 0.05 * holes.size, (holes.size, lines, runs)


def test_tools_postorder_rejects_what_is_not_a_forest():
    from river_route_amd import tools
    with pytest.raises(ValueError, match='Unknown downstream_river_id: 99'):
        tools.postorder(np.array([1, 2, 3]), np.array([2, 99, -1]))
    with pytest.raises(ValueError, match='not a forest'):
        tools.postorder(np.array([1, 2, 3, 4]), np.array([2, 3, 1, -1]))      # 1 -> 2 -> 3 -> 1
    with pytest.raises(ValueError, match='not a forest'):
        tools.postorder(np.array([7]), np.array([7]))
    assert tools.postorder(np.array([], dtype=np.int64), np.array([], dtype=np.int64)).size == 0


@pytest.mark.parametrize('order', ['random', 'levels', 'bfs'])
def test_other_orders_keep_to_records(order):
    net = synth.synth_network(20_000, seed=3, order=order)
    indptr, indices = csc_from_down(net.down_index)
    with Plan(indptr, indices, device=RR_DEVICE_NONE) as plan:
        info = plan.direct_info()
    assert not info['ok'] and 'contiguous' in info['why']


def test_chains_and_wide_confluences_keep_to_records():
    n = 3000
    chain = np.arange(1, n + 1, dtype=np.int64)
    chain[-1] = -1
    indptr, indices = csc_from_down(chain)
    with Plan(indptr, indices, device=RR_DEVICE_NONE) as plan:      # a chain is contiguous in any order, but every subtree taller than the window is skeleton
        info = plan.direct_info()
    assert not info['ok'] and 'tenth' in info['why']
    fan = np.concatenate([np.full(5, 5), [-1]]).astype(np.int64)      # five tributaries into one reach
    indptr, indices = csc_from_down(fan)
    with Plan(indptr, indices, device=RR_DEVICE_NONE) as plan:
        info = plan.direct_info()
    assert not info['ok'] and 'three' in info['why']


# ---------------------------------------------------------------------------------------------- on the GPU box

def _case(n, seed, order='postorder', chainy=False):
    net = (synth.synth_network_chain(n, p_chain=0.3, n_outlets=5, p_third=0.04, seed=seed, order=order) if chainy
           else synth.synth_network(n, seed=seed, order=order))
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0)
    return net, indptr, indices, c1, c2, c3


@pytest.mark.gpu
@pytest.mark.parametrize('n,T,env,chainy', [(9, 40, {}, False), (1000, 100, {}, False), (60_000, 200, {'RR_WAVE_K': '64'}, False),
                                             (60_000, 333, {'RR_WAVE_K': '128'}, True), (300_000, 150, {}, False),
                                             (120_000, 600, {'RR_WAVE_K': '256'}, False), (60_000, 97, {'RR_TILE_BLOCK': '64', 'RR_WAVE_K': '32'}, False),
                                             (60_000, 1500, {'RR_WAVE_K': '1024'}, False)])      # lane tasks of 1,024 rows (the second one partial) over skeleton tasks of 128 ticks
def test_direct_rows_vs_oracle(monkeypatch, n, T, env, chainy):
    """rr_rapid_route_dev on a post-order network: the direct row path runs (plan.last_kernel()), two consecutive calls (state
    carried), non-zero initial state -- discharge rows and state against the oracle."""
    set_env(monkeypatch, env)
    net, indptr, indices, c1, c2, c3 = _case(n, 17, chainy=chainy)
    lhs, c4_dt = -c1[indices], (c1 + c2) / 900.0
    q0 = 2.0 * synth.u01(3, np.arange(n))
    with Plan(indptr, indices) as plan:
        assert plan.direct_info()['ok'], plan.direct_info()['why']
        plan.set_coeffs(lhs, c2, c3, c4_dt)
        d_q, d_ql, d_out = DeviceBuffer(n * 8).upload(q0), DeviceBuffer(T * n * 8), DeviceBuffer(T * n * 8)
        q_ref = q0.copy()
        for call in range(2):
            ql = synth.synth_qlateral(n, call * T, (call + 1) * T)
            d_ref = np.zeros((T, n))
            oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, ql, d_ref, 1)
            d_ql.upload(ql)
            plan.rapid_route_dev(d_q, d_ql, T, d_out, T, T, 1)
            assert plan.last_kernel() == 'direct'
            assert_close(d_out.download(np.float64, (T, n)), d_ref, f'discharge, call {call}')
            assert_close(d_q.download(np.float64, (n,)), q_ref, f'q_t, call {call}')
        for b in (d_q, d_ql, d_out):
            b.free()


@pytest.mark.gpu
def test_direct_rows_equal_the_record_path_bit_for_bit(monkeypatch):
    """The same call through the direct row path and through k_tile + the record passes (RR_DIRECT=0): the two add a reach's
    upstream discharges in the same order and evaluate the same fused multiply-adds, so every row agrees to the last bit."""
    n, T = 200_000, 300
    net, indptr, indices, c1, c2, c3 = _case(n, 23)
    ql = synth.synth_qlateral(n, 0, T)
    q0 = synth.u01(9, np.arange(n))
    res = {}
    for direct in ('1', '0'):
        set_env(monkeypatch, {'RR_DIRECT': direct})
        with Plan(indptr, indices) as plan:
            plan.set_coeffs(-c1[indices], c2, c3, (c1 + c2) / 900.0)
            d_q, d_ql, d_out = DeviceBuffer(n * 8).upload(q0), DeviceBuffer(ql.nbytes).upload(ql), DeviceBuffer(T * n * 8)
            plan.rapid_route_dev(d_q, d_ql, T, d_out, T, T, 1)
            assert plan.last_kernel() == ('direct' if direct == '1' else 'tile')
            res[direct] = (d_out.download(np.float64, (T, n)), d_q.download(np.float64, (n,)))
            for b in (d_q, d_ql, d_out):
                b.free()
    np.testing.assert_array_equal(res['1'][0], res['0'][0])
    np.testing.assert_array_equal(res['1'][1], res['0'][1])


@pytest.mark.gpu
def test_direct_rows_cyclic_forcing_and_ring_goes_round_vs_oracle(monkeypatch):
    """A long call on a cyclic forcing array (row t % rows, what bench.py feeds): the skeleton's record ring goes round four
    times (RR_WAVE_K=64 keeps it short), 118 direct launches -- every row against the oracle fed the same rows."""
    set_env(monkeypatch, {'RR_WAVE_K': '64'})
    n, T, rows = 60_000, 7500, 75
    net, indptr, indices, c1, c2, c3 = _case(n, 29)
    lhs, c4_dt = -c1[indices], (c1 + c2) / 900.0
    ql = synth.synth_qlateral(n, 0, rows)
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(lhs, c2, c3, c4_dt)
        sched = plan.reserve(0, T, 1)
        assert sched['direct'] and sched['ring_chunks'] * 16 * 3 < T, sched
        d_q, d_ql, d_out = DeviceBuffer(n * 8).upload(np.zeros(n)), DeviceBuffer(ql.nbytes).upload(ql), DeviceBuffer(T * n * 8)
        plan.rapid_route_dev(d_q, d_ql, rows, d_out, T, T, 1)
        assert plan.last_kernel() == 'direct'
        got, q_got = d_out.download(np.float64, (T, n)), d_q.download(np.float64, (n,))
        for b in (d_q, d_ql, d_out):
            b.free()
    q_ref = np.zeros(n)
    for r0 in range(0, T, rows):      # the oracle over the same cyclic rows, block by block
        d_ref = np.zeros((rows, n))
        oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, ql, d_ref, 1)
        assert_close(got[r0:r0 + rows], d_ref, f'rows {r0}..')
    assert_close(q_got, q_ref, 'q_t')


@pytest.mark.gpu
def test_direct_rows_at_1m_reaches_vs_oracle():
    """BASELINE config 3's network in post-order, 640 rows in tasks of 256 over skeleton tasks of 128 ticks (the schedule's choice for a call of that length): 4,938
    column-range tiles, 50,021 holes patched from the skeleton's 30 tile levels -- all 1M columns of every row against the oracle."""
    n, T = 1_000_000, 640
    net, indptr, indices, c1, c2, c3 = _case(n, synth.NETWORK_SEED)
    lhs, c4_dt = -c1[indices], (c1 + c2) / 900.0
    ql = synth.synth_qlateral(n, 0, T)
    q_ref, d_ref = np.zeros(n), np.zeros((T, n))
    oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, ql, d_ref, 1)
    with Plan(indptr, indices) as plan:
        info = plan.direct_info()
        assert info['ok'] and info['tiles'] > 4000 and info['skeleton_levels'] > 10
        plan.set_coeffs(lhs, c2, c3, c4_dt)
        d_q, d_ql, d_out = DeviceBuffer(n * 8).upload(np.zeros(n)), DeviceBuffer(ql.nbytes).upload(ql), DeviceBuffer(T * n * 8)
        plan.rapid_route_dev(d_q, d_ql, T, d_out, T, T, 1)
        assert plan.last_kernel() == 'direct' and plan.profile()['ticks_per_launch'] == 256
        assert_close(d_out.download(np.float64, (T, n)), d_ref, 'discharge')
        assert_close(d_q.download(np.float64, (n,)), q_ref, 'q_t')
        for b in (d_q, d_ql, d_out):
            b.free()


@pytest.mark.gpu
def test_calls_the_direct_path_does_not_take_fall_back_to_records(monkeypatch):
    """On a post-order plan: more sub-steps than the direct path takes (five) keep to the record path (against the oracle), float32 rows
    out take the direct path since round 5, and the plan goes back and forth between the two."""
    set_env(monkeypatch, {})
    n, T = 50_000, 64
    net, indptr, indices, c1, c2, c3 = _case(n, 31)
    c1s, c2s, c3s = oracle.muskingum_coefficients(net.k, net.x, 180.0)
    ql = synth.synth_qlateral(n, 0, T)
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(-c1s[indices], c2s, c3s, (c1s + c2s) / 900.0)
        q_ref, d_ref = np.zeros(n), np.zeros((T, n))
        oracle.rapid_route(indptr, indices, -c1s[indices], c2s, c3s, (c1s + c2s) / 900.0, q_ref, ql, d_ref, 5)
        d_q, d_ql, d_out = DeviceBuffer(n * 8).upload(np.zeros(n)), DeviceBuffer(ql.nbytes).upload(ql), DeviceBuffer(T * n * 8)
        plan.rapid_route_dev(d_q, d_ql, T, d_out, T, T, 5)
        assert plan.last_kernel() == 'tile'
        assert_close(d_out.download(np.float64, (T, n)), d_ref, 'five sub-steps')
        plan.set_coeffs(-c1[indices], c2, c3, (c1 + c2) / 900.0)
        q_ref, d_ref = np.zeros(n), np.zeros((T, n))
        oracle.rapid_route(indptr, indices, -c1[indices], c2, c3, (c1 + c2) / 900.0, q_ref, ql, d_ref, 1)
        d_ql32, d_out32 = DeviceBuffer(T * n * 4).upload(ql.astype(np.float32)), DeviceBuffer(T * n * 4)
        d_q.upload(np.zeros(n))
        plan.rapid_route_f32_dev(d_q, d_ql, T, d_out32, T, 1, 1)
        assert plan.last_kernel() == 'direct'
        np.testing.assert_allclose(d_out32.download(np.float32, (T, n)), d_ref.astype(np.float32), rtol=1.2e-7, atol=1e-10 * np.abs(d_ref).max())
        d_q.upload(np.zeros(n))
        plan.rapid_route_dev(d_q, d_ql, T, d_out, T, T, 1)      # and back on the direct path, after the record path used the plan
        assert plan.last_kernel() == 'direct'
        assert_close(d_out.download(np.float64, (T, n)), d_ref, 'direct after records')
        for b in (d_q, d_ql, d_out, d_ql32, d_out32):
            b.free()


@pytest.mark.gpu
def test_direct_full_year_at_1m_sub_basins_vs_oracle(monkeypatch):
    """The post-order line of the bench at full length against the ORACLE (the bench's own gate checks 96 rows): four sub-basins of
    3k-6k reaches -- each with skeleton reaches, holes and outlets that feed them -- routed by the oracle on their own through
    all 35,040 steps; the engine's rows for those columns, from two calls of 17,520 rows over the whole 1M-reach network in tasks of
    1,024 rows over skeleton tasks of 512 ticks (18 launches of k_direct per call, the skeleton's record ring, state carried between the calls, a 120-row forcing
    ring read 146 times per call) must be theirs row by row."""
    import torch
    from test_gpu_tiles import _sub_basins
    set_env(monkeypatch, {})
    n, T, rows, calls = 1_000_000, 35_040, 120, 2
    net, indptr, indices, c1, c2, c3 = _case(n, synth.NETWORK_SEED)
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
        info = plan.direct_info()
        assert info['ok'] and info['holes'] > 10_000
        L = plan.direct_layout()
        assert (L['delay'][cols] & HOLE).any(), 'the sub-basins hold skeleton reaches'
        plan.set_coeffs(lhs, c2, c3, c4_dt)
        q = torch.zeros(n, dtype=torch.float64, device=dev)
        out = torch.empty((Tc, n), dtype=torch.float64, device=dev)
        for call in range(calls):
            out.fill_(-1.0)
            plan.rapid_route_dev(q, ql, rows, out, Tc, Tc, 1, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert plan.last_kernel() == 'direct' and plan.profile()['ticks_per_launch'] == 1024
            kept = out[:, cols_t]
            for r0 in range(0, Tc, rows):
                oracle.rapid_route(s_indptr, s_indices, s_lhs, s_c2, s_c3, s_c4, q_ref, ql_sub, d_ref, 1)
                got = kept[r0:r0 + rows].cpu().numpy()
                err = np.abs(got - d_ref).max() / np.abs(d_ref).max()
                worst = max(worst, err)
                assert err <= 1e-10, f'call {call}, rows {r0}..{r0 + rows}: {err:.3e} of the largest discharge'
            del kept
        assert_close(q[cols_t].cpu().numpy(), q_ref, 'final state of the sub-basins')
    del out, q, ql, cols_t
    torch.cuda.empty_cache()      # 140 GB of discharge rows: handed back, or the engine of a later test sees a card too full for its record ring
    print(f'post-order sub-basins: {cols.size} reaches x {T} steps against the oracle, worst difference {worst:.2e} of the largest discharge')


@pytest.mark.gpu
def test_direct_constant_forcing_settles_at_the_basin_sums(monkeypatch):
    """The fixed-point property of test_gpu_tiles.py at full size on the direct row path: one forcing row (a ring of ONE row: the
    rows-in waves read it 35,040 times) and a 128-row sink written 274 times; after a year every one of the 1M reaches -- small
    subtrees, holes, skeleton -- sits at the lateral inflow accumulated over its basin (_numba_kernels.py:68-78: q = A q + ql/dt
    is the update's fixed point)."""
    import torch
    set_env(monkeypatch, {})
    n, T, sink = 1_000_000, 35_040, 128
    net, indptr, indices, c1, c2, c3 = _case(n, synth.NETWORK_SEED)
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
        assert plan.last_kernel() == 'direct'
        np.testing.assert_allclose(q.cpu().numpy(), want, rtol=1e-9, err_msg='final state')
        np.testing.assert_allclose(out.cpu().numpy(), np.broadcast_to(want, (sink, n)), rtol=1e-9, err_msg='last 128 rows')
    del out, q
    torch.cuda.empty_cache()


@pytest.mark.gpu
def test_stream_session_takes_the_direct_path_vs_oracle(monkeypatch):
    """rr_stream_begin / advance / end on a post-order network: HipPartEngine hands over ghost and export buffers even
    where the part has no boundary reach (one part = the whole network) -- the call takes the direct row path and reproduces the oracle."""
    import torch
    from river_route_amd.engine import partition_forest
    from river_route_amd.multi_gpu import HipPartEngine, split_network
    set_env(monkeypatch, {})
    n, T = 80_000, 200
    net, indptr, indices, c1, c2, c3 = _case(n, 41)
    lhs, c4_dt = -c1[indices], (c1 + c2) / 900.0
    q0 = synth.u01(4, np.arange(n))
    ql = synth.synth_qlateral(n, 0, T)
    q_ref, d_ref = q0.copy(), np.zeros((T, n))
    oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, ql, d_ref, 1)
    part_of, _ = partition_forest(indptr, indices, 1)
    spec = split_network(net.down_index, part_of, 0, 1)
    eng = HipPartEngine(spec, c1, c2, c3, c4_dt, q0, ql, T, 1, 0, out_rows=T)
    eng.begin()
    for rows in (64, 130, T):      # rows announced in three steps
        eng.advance(rows, rows)
    eng.end()
    torch.cuda.synchronize()
    assert eng.plan.last_kernel() == 'direct'
    assert_close(eng.discharge.cpu().numpy(), d_ref, 'discharge')
    assert_close(eng.final_state(), q_ref, 'state')
    eng.close()


@pytest.mark.gpu
def test_stream_session_with_a_refilled_lateral_ring_shorter_than_a_task(monkeypatch):
    """A caller that REFILLS a 48-row cyclic lateral ring between rr_stream_advance calls (step t reads row t % 48): the direct
    task is capped to the ring (K = 48 instead of 64; 32 for a ring of 40 rows), so no launch reads a slot that has been refilled.
    Every row against the oracle."""
    import torch
    set_env(monkeypatch, {})
    n, T = 60_000, 300
    net, indptr, indices, c1, c2, c3 = _case(n, 43)
    lhs, c4_dt = -c1[indices], (c1 + c2) / 900.0
    ql = synth.synth_qlateral(n, 0, T)
    q_ref, d_ref = np.zeros(n), np.zeros((T, n))
    oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, ql, d_ref, 1)
    dev = torch.device('cuda:0')
    for ring, K in ((48, 48), (40, 32)):
        with Plan(indptr, indices) as plan:
            plan.set_coeffs(lhs, c2, c3, c4_dt)
            q = torch.zeros(n, dtype=torch.float64, device=dev)
            lat = torch.zeros((ring, n), dtype=torch.float64, device=dev)
            out = torch.zeros((T, n), dtype=torch.float64, device=dev)
            stream = torch.cuda.current_stream().cuda_stream
            plan.stream_begin(q, lat, ring, out, T, T, 1, stream=stream)
            fed = 0
            while fed < T:      # a whole task at a time: what is announced is taken at once, and never more than the ring holds
                step = min(K, T - fed)
                rows = torch.from_numpy(ql[fed:fed + step]).to(dev)
                idx = torch.arange(fed, fed + step, device=dev) % ring
                lat[idx] = rows      # on the call's stream: ordered behind the launches that read the old rows
                fed += step
                plan.stream_advance(fed, fed)
            plan.stream_end(q)
            torch.cuda.synchronize()
            assert plan.last_kernel() == 'direct' and plan.profile()['ticks_per_launch'] == K
            assert_close(out.cpu().numpy(), d_ref, f'discharge, ring of {ring} rows')
            assert_close(q.cpu().numpy(), q_ref, 'state')


def test_parts_of_a_cut_postorder_network_lay_out_around_their_boundary_reaches():
    """rr_plan_set_boundary lays the direct plan out again: a ghost's column (first in a part's local order, far from the reach it
    flows into) is passed through, the reaches below it join the skeleton, an export is a lane's or the skeleton's."""
    from river_route_amd.engine import partition_forest
    from river_route_amd.multi_gpu import split_network
    n, parts = 400_000, 4
    net = synth.synth_network(n, seed=4, order='postorder')
    indptr, indices = csc_from_down(net.down_index)
    part_of, _ = partition_forest(indptr, indices, parts)
    for p in (0, parts - 1):
        spec = split_network(net.down_index, part_of, p, parts)
        with Plan(spec.indptr, spec.indices, device=RR_DEVICE_NONE) as plan:
            export_local = spec.n_ghost + np.searchsorted(spec.real_global, spec.export_global)
            plan.set_boundary(np.arange(spec.n_ghost), export_local)
            info = plan.direct_info()
            assert info['ok'], info['why']
            L = plan.direct_layout()
        word = L['delay'].astype(np.int64)
        assert np.all((word[:spec.n_ghost] & HOLE) != 0) and np.all(((word[:spec.n_ghost] >> 8) & 0x7F) == 0), 'a ghost is passed through and sends nothing'
        below = spec.down_local[:spec.n_ghost]
        assert np.all((word[below] & HOLE) != 0) and np.all(L['xinfo'][below] >= 0), 'the reach a ghost flows into is a skeleton reach'
        lane_export = (word[export_local] & (1 << 15)) != 0
        assert np.array_equal(L['xinfo'][export_local][lane_export], np.flatnonzero(lane_export)), 'a lane export carries its column of the export series'
        assert np.all((word[export_local][~lane_export] & HOLE) != 0), 'the other exports are skeleton reaches'
        if p == 0:
            assert lane_export.any() and not lane_export.all(), 'this case has both kinds'


def _route_parts_on_the_direct_path(n, parts, T, chunk, seed, expect_direct, chainy=False):
    from river_route_amd.engine import partition_forest
    from river_route_amd.multi_gpu import HipPartEngine, run_in_process, split_network
    net, indptr, indices, c1, c2, c3 = _case(n, seed, chainy=chainy)
    c4_dt = (c1 + c2) / 900.0
    q0 = 4.0 * synth.u01(8, np.arange(n))
    ql = synth.synth_qlateral(n, 0, T)
    q_ref, d_ref = q0.copy(), np.zeros((T, n))
    oracle.rapid_route(indptr, indices, -c1[indices], c2, c3, c4_dt, q_ref, ql, d_ref, 1)
    part_of, _ = partition_forest(indptr, indices, parts)
    specs = [split_network(net.down_index, part_of, p, parts) for p in range(parts)]
    engines = [HipPartEngine(s, c1, c2, c3, c4_dt, q0, ql[:, s.real_global], T, 1, 0, out_rows=T) for s in specs]
    run_in_process(engines, specs, T, 1, chunk)
    kernels = [e.plan.last_kernel() for e in engines]
    q, d = np.zeros(n), np.zeros((T, n))
    for s, e in zip(specs, engines):
        q[s.real_global] = e.final_state()
        d[:, s.real_global] = e.discharge.cpu().numpy()[:, s.n_ghost:]
        e.close()
    assert_close(q, q_ref, 'state')
    assert_close(d, d_ref, 'discharge')
    assert kernels == expect_direct, kernels
    return specs


@pytest.mark.gpu
def test_partitioned_postorder_network_on_the_direct_path_vs_oracle(monkeypatch):
    """Parts of a cut post-order network through rr_stream_begin / advance / end with their boundary series exchanged in batches
    (multi_gpu.run_in_process: the distributed driver without the network): the leaf parts' exports -- lanes' and skeleton reaches' --
    feed the trunk part's ghosts, every part on the direct row path, against the oracle on the undivided network.  At 200k reaches
    the trunk part is mostly main stem and keeps to records (its ghosts then go through the record path's in-pass)."""
    set_env(monkeypatch, {})
    specs = _route_parts_on_the_direct_path(1_000_000, 4, 300, 64, 4, ['direct'] * 4)
    assert specs[-1].n_ghost > 100
    _route_parts_on_the_direct_path(200_000, 8, 200, 32, 4, ['direct'] * 7 + ['tile'])
    set_env(monkeypatch, {'RR_WAVE_K': '64'})      # short tasks: the skeleton's ring goes round, exports trail by fewer rows
    _route_parts_on_the_direct_path(400_000, 4, 700, 48, 6, ['direct'] * 4)
    set_env(monkeypatch, {'RR_WAVE_K': '1024'})    # the year's lane tasks (1,024 rows, the second one partial) over the 64-tick skeleton tasks of a part with exports: 16 skeleton launches per direct launch
    _route_parts_on_the_direct_path(400_000, 4, 1300, 128, 6, ['direct'] * 4)


@pytest.mark.gpu
@pytest.mark.parametrize('n,parts,T,chunk,seed,chainy', [(20_000, 5, 40, 40, 812328, True), (20_000, 5, 130, 64, 812328, True), (3000, 7, 200, 32, 362543, False),
                                                         (3000, 7, 300, 32, 362543, False)])
def test_short_calls_of_shallow_parts_on_the_direct_path_vs_oracle(monkeypatch, n, parts, T, chunk, seed, chainy):
    """Regression (found by profiles/microbench/parts_fuzz.py): the skeleton's record ring is cut to the call's length, and with boundary ghosts it has to
    hold their in-pass's batches whole -- nine records per position and batch, zeros past the call's end.  In a shorter ring a batch wrapped onto its own
    first records: a 40- or 130-row call of a shallow part of a chain-grown network routed wrong boundary inflow, a 200-row call of a 3,000-reach network
    in seven parts never finished."""
    set_env(monkeypatch, {})
    specs = _route_parts_on_the_direct_path(n, parts, T, chunk, seed, ['direct'] * parts, chainy=chainy)
    assert sum(s.n_ghost for s in specs) > 0


@pytest.mark.gpu
@pytest.mark.parametrize('n,T,factor,in32,env', [(60_000, 256, 1, False, {}), (60_000, 256, 4, True, {}), (120_001, 640, 8, True, {}), (60_000, 384, 128, False, {}),
                                                 (60_000, 200, 1, True, {'RR_WAVE_K': '128'}), (300_000, 1024, 4, True, {})])
def test_float32_rows_on_the_direct_path(monkeypatch, n, T, factor, in32, env):
    """What the routers call for a qlateral file (TransformMuskingum.py:128-142 fused in): float32 lateral rows in and / or float32 rows
    out, each the mean of `factor` routed rows, on the direct row path -- equal BIT FOR BIT to the record path (RR_DIRECT=0) and to the
    float64 direct call followed by numpy's mean and cast; against the oracle to 1 ulp(float32)."""
    net, indptr, indices, c1, c2, c3 = _case(n, 51)
    lhs, c4_dt = -c1[indices], (c1 + c2) / 900.0
    ql = synth.synth_qlateral(n, 0, T)
    if in32:
        ql = ql.astype(np.float32).astype(np.float64)      # the float32 file's values
    q0 = synth.u01(2, np.arange(n))
    q_ref, d_ref = q0.copy(), np.zeros((T, n))
    oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, ql, d_ref, 1)
    want = d_ref.reshape(T // factor, factor, n).mean(axis=1).astype(np.float32)
    res = {}
    for direct in ('1', '0'):
        set_env(monkeypatch, dict(env, RR_DIRECT=direct))
        with Plan(indptr, indices) as plan:
            plan.set_coeffs(lhs, c2, c3, c4_dt)
            d_q = DeviceBuffer(n * 8).upload(q0)
            d_out32 = DeviceBuffer((T // factor) * n * 4)
            if in32:
                d_ql = DeviceBuffer(T * n * 4).upload(ql.astype(np.float32))
                plan.rapid_route_f32in_dev(d_q, d_ql, T, T, 1, discharge32=d_out32, factor=factor)
            else:
                d_ql = DeviceBuffer(T * n * 8).upload(ql)
                plan.rapid_route_f32_dev(d_q, d_ql, T, d_out32, T, 1, factor)
            assert plan.last_kernel() == ('direct' if direct == '1' else 'tile')
            res[direct] = (d_out32.download(np.float32, (T // factor, n)), d_q.download(np.float64, (n,)))
            if direct == '1':      # the float64 rows of the same plan, then numpy
                d_q.upload(q0)
                d_out = DeviceBuffer(T * n * 8)
                if in32:
                    plan.rapid_route_f32in_dev(d_q, d_ql, T, T, 1, discharge=d_out, out_rows=T)
                else:
                    plan.rapid_route_dev(d_q, d_ql, T, d_out, T, T, 1)
                assert plan.last_kernel() == 'direct'
                full = d_out.download(np.float64, (T, n))
                d_out.free()
            for b in (d_q, d_ql, d_out32):
                b.free()
    np.testing.assert_array_equal(res['1'][0], res['0'][0])
    np.testing.assert_array_equal(res['1'][1], res['0'][1])
    np.testing.assert_array_equal(res['1'][0], full.reshape(T // factor, factor, n).mean(axis=1).astype(np.float32) if factor > 1 else full.astype(np.float32))
    assert_close(full, d_ref, 'float64 rows')
    np.testing.assert_allclose(res['1'][0], want, rtol=1.2e-7, atol=1e-10 * float(np.abs(want).max()))
    assert_close(res['1'][1], q_ref, 'state')


@pytest.mark.gpu
@pytest.mark.parametrize('n,T,nsub,mode,env', [(60_000, 100, 2, 'rapid', {}), (60_000, 70, 4, 'rapid', {}), (200_000, 130, 3, 'rapid', {'RR_WAVE_K': '64'}),
                                               (60_000, 96, 1, 'muskingum', {}), (60_000, 60, 4, 'muskingum', {}), (1_000_000, 80, 4, 'rapid', {}),
                                               (120_000, 300, 2, 'rapid', {'RR_WAVE_K': '32'}),
                                               (20_000, 1100, 2, 'rapid', {'RR_WAVE_K': '1024'}), (20_000, 1100, 1, 'muskingum', {'RR_WAVE_K': '1024'})])      # the year's task length: 1,024 rows over shorter skeleton tasks
def test_substeps_and_channel_only_routing_on_the_direct_path(monkeypatch, n, T, nsub, mode, env):
    """Routing sub-steps (dt_routing < dt_runoff: _numba_kernels.py:66-84, the row's lateral value held, the output the mean of the
    sub-steps) and channel-only routing (Muskingum.py:262-290) on the direct row path: two consecutive calls against the oracle, and
    bit for bit the record path's rows (RR_DIRECT=0)."""
    net, indptr, indices, c1, c2, c3 = _case(n, 61)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0 / nsub)
    lhs, c4_dt = -c1[indices], (c1 + c2) / 900.0
    q0 = 3.0 * synth.u01(6, np.arange(n))
    res = {}
    for direct in ('1', '0'):
        set_env(monkeypatch, dict(env, RR_DIRECT=direct))
        with Plan(indptr, indices) as plan:
            plan.set_coeffs(lhs, c2, c3, c4_dt if mode == 'rapid' else None)
            d_q, d_ql, d_out = DeviceBuffer(n * 8).upload(q0), DeviceBuffer(T * n * 8), DeviceBuffer(T * n * 8)
            q_ref, rows = q0.copy(), []
            for call in range(2):
                d_ref = np.zeros((T, n))
                if mode == 'rapid':
                    ql = synth.synth_qlateral(n, call * T, (call + 1) * T)
                    oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, ql, d_ref, nsub)
                    d_ql.upload(ql)
                    plan.rapid_route_dev(d_q, d_ql, T, d_out, T, T, nsub)
                else:
                    oracle.muskingum_route(indptr, indices, lhs, c2, c3, q_ref, d_ref, T, nsub)
                    plan.muskingum_route_dev(d_q, d_out, T, T, nsub)
                assert plan.last_kernel() == ('direct' if direct == '1' else 'tile')
                got = d_out.download(np.float64, (T, n))
                assert_close(got, d_ref, f'discharge, call {call}')
                assert_close(d_q.download(np.float64, (n,)), q_ref, f'q_t, call {call}')
                rows.append(got)
            res[direct] = rows
            for b in (d_q, d_ql, d_out):
                b.free()
    for a, b in zip(res['1'], res['0']):
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize('n,T,env,f32,nsub', [(9, 40, {}, False, 1), (1000, 100, {}, False, 1), (60_000, 200, {'RR_WAVE_K': '64'}, False, 1), (300_000, 150, {}, False, 1),
                                              (120_000, 600, {'RR_WAVE_K': '256'}, False, 1), (60_000, 256, {}, True, 1),
                                              (60_000, 100, {}, False, 2), (120_000, 130, {'RR_WAVE_K': '64'}, False, 3), (60_000, 128, {}, True, 4), (1000, 70, {}, False, 4)])      # sub-steps
def test_unit_route_on_the_direct_path_vs_oracle(monkeypatch, n, T, env, f32, nsub):
    """rr_unit_route_dev (rows of convolved lateral inflow) on a post-order network: k_direct<UNIT> routes the small sub-basins, the skeleton
    runs k_tile<UNIT> on records; two files with the state hand-off of UnitMuskingum._router (river_route/routers/UnitMuskingum.py:72-98);
    headwater columns leave unclipped (_numba_kernels.py:122-123).  f32: rr_unit_route_f32_dev, the routers' float32 means of 4 rows."""
    from tests_support import unit_split_arrays
    set_env(monkeypatch, env)
    net, indptr, indices, c1, c2, c3 = _case(n, 31)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0 / nsub)
    hw_idx, inner_idx, A_in, A_hw = unit_split_arrays(indptr, indices, n)
    c1i, c2i, c3i = c1[inner_idx], c2[inner_idx], c3[inner_idx]
    args = (A_in.indptr, A_in.indices, -c1i[A_in.indices], A_in.indptr, A_in.indices, A_in.data,
            A_hw.indptr, A_hw.indices, A_hw.data, c1i, c2i, c3i, hw_idx, inner_idx)
    uh = oracle.UnitHydrograph(synth.synth_uh_kernel(n, 12))
    state_ref = 3.0 * synth.u01(7, np.arange(n))
    state = state_ref.copy()
    ni = inner_idx.size
    factor = 4
    with Plan(indptr, indices) as plan:
        assert plan.direct_info()['ok'], plan.direct_info()['why']
        plan.set_coeffs(-c1[indices], c2, c3, None)
        d_conv, d_out = DeviceBuffer(T * n * 8), DeviceBuffer(T * n * 8)
        d_qc, d_qf = DeviceBuffer(max(ni, 1) * 8), DeviceBuffer(max(ni, 1) * 8)
        for f in range(2):
            conv_ref = uh.convolve(synth.synth_runoff_depth(n, f * T, (f + 1) * T))
            if f == 1:
                conv_ref[5, hw_idx[:3]] = -0.25      # a negative headwater inflow stays as it is
            qc_ref, qf_ref, d_ref = state_ref[inner_idx].copy(), state_ref[inner_idx].copy(), np.zeros((T, n))
            oracle.unit_route(*args, qc_ref, qf_ref, conv_ref, d_ref, nsub)
            state_ref[hw_idx], state_ref[inner_idx] = conv_ref[-1][hw_idx], qf_ref
            d_conv.upload(conv_ref)
            d_qc.upload(state[inner_idx].copy()); d_qf.upload(state[inner_idx].copy())
            if f32:
                plan.unit_route_f32_dev(d_qc, d_qf, d_conv, T, d_out, T, nsub, factor=factor)
            else:
                plan.unit_route_dev(d_qc, d_qf, d_conv, T, d_out, T, T, nsub)
            assert plan.last_kernel() == 'direct'
            qc, qf = d_qc.download(np.float64, (ni,)), d_qf.download(np.float64, (ni,))
            state[hw_idx], state[inner_idx] = conv_ref[-1][hw_idx], qf
            assert_close(qc, qc_ref, f'file {f} q_ch')
            assert_close(qf, qf_ref, f'file {f} q_full')
            if f32:
                want = d_ref.reshape(T // factor, factor, n).mean(axis=1).astype(np.float32)
                got = d_out.download(np.float32, (T // factor, n))
                assert np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64)).max() <= 1, 'float32 means differ by more than 1 ulp'
            else:
                d = d_out.download(np.float64, (T, n))
                assert_close(d, d_ref, f'file {f} discharge')
                np.testing.assert_array_equal(d[:, hw_idx], conv_ref[:, hw_idx])     # headwaters: unclamped, un-averaged
        for b in (d_conv, d_out, d_qc, d_qf):
            b.free()


@pytest.mark.gpu
@pytest.mark.parametrize('n,T,n_ks,in32,f32out,nsub', [(60_000, 200, 48, False, False, 1), (300_000, 150, 48, True, False, 1), (60_000, 256, 12, True, True, 1), (60_007, 300, 33, False, False, 1),
                                                        (60_000, 100, 60, False, False, 1), (60_000, 96, 24, True, False, 3), (60_000, 128, 48, False, True, 2)])      # sub-steps
def test_unit_route_with_convolution_on_the_direct_path_vs_oracle(monkeypatch, n, T, n_ks, in32, f32out, nsub):
    """rr_unit_route_uh_dev / rr_unit_route_uh_f32in_dev on a post-order network: UnitHydrograph.convolve + unit_route + the router's state
    bookkeeping (river_route/routers/UnitMuskingum.py:72-98) in one call -- on the direct row path the convolution runs as a pass of its own into
    work rows, then k_direct<UNIT> routes them; two files, the second shorter than the kernel, so the carry-over state crosses a file boundary."""
    from tests_support import unit_split_arrays
    set_env(monkeypatch, {})
    net, indptr, indices, c1, c2, c3 = _case(n, 23)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0 / nsub)
    hw_idx, inner_idx, A_in, A_hw = unit_split_arrays(indptr, indices, n)
    c1i, c2i, c3i = c1[inner_idx], c2[inner_idx], c3[inner_idx]
    args = (A_in.indptr, A_in.indices, -c1i[A_in.indices], A_in.indptr, A_in.indices, A_in.data,
            A_hw.indptr, A_hw.indices, A_hw.data, c1i, c2i, c3i, hw_idx, inner_idx)
    kern = synth.synth_uh_kernel(n, n_ks)
    uh = oracle.UnitHydrograph(kern)
    state_ref = 3.0 * synth.u01(7, np.arange(n))
    ni = inner_idx.size
    factor = 4
    with Plan(indptr, indices) as plan:
        assert plan.direct_info()['ok'], plan.direct_info()['why']
        plan.set_coeffs(-c1[indices], c2, c3, None)
        d_kern, d_state = DeviceBuffer(kern.nbytes).upload(kern), DeviceBuffer(kern.nbytes).upload(np.zeros_like(kern))
        d_depth, d_out, d_fin = DeviceBuffer(T * n * 8), DeviceBuffer(T * n * 8), DeviceBuffer(n * 8)
        d_qc, d_qf = DeviceBuffer(ni * 8), DeviceBuffer(ni * 8)
        state = state_ref.copy()
        for f, Tf in enumerate((T, max(8, n_ks // 2 // factor * factor))):
            depth = synth.synth_runoff_depth(n, f * T, f * T + Tf)
            if in32:
                depth = depth.astype(np.float32)
            conv_ref = uh.convolve(depth.astype(np.float64))
            qc_ref, qf_ref, d_ref = state_ref[inner_idx].copy(), state_ref[inner_idx].copy(), np.zeros((Tf, n))
            oracle.unit_route(*args, qc_ref, qf_ref, conv_ref, d_ref, nsub)
            state_ref[hw_idx], state_ref[inner_idx] = conv_ref[-1][hw_idx], qf_ref
            d_depth.upload(depth)
            d_qc.upload(state[inner_idx].copy()); d_qf.upload(state[inner_idx].copy())
            call = plan.unit_route_uh_f32in_dev if in32 else plan.unit_route_uh_dev
            out_kw = dict(discharge32=d_out, factor=factor) if f32out else dict(discharge=d_out)
            call(d_qc, d_qf, d_fin, d_kern, d_state, n_ks, d_depth, Tf, nsub, **out_kw)
            assert plan.last_kernel() == 'direct'
            state = d_fin.download(np.float64, (n,))
            if f32out:
                want = d_ref.reshape(Tf // factor, factor, n).mean(axis=1).astype(np.float32)
                got = d_out.download(np.float32, (Tf // factor, n))
                assert np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64)).max() <= 1, 'float32 means differ by more than 1 ulp'
            else:
                assert_close(d_out.download(np.float64, (Tf, n)), d_ref, f'file {f} discharge')
            assert_close(state, state_ref, f'file {f} router state')
            assert_close(d_qc.download(np.float64, (ni,)), qc_ref, f'file {f} q_ch')
            assert_close(d_state.download(np.float64, kern.shape), uh.state, f'file {f} UH state')
        for b in (d_kern, d_state, d_depth, d_out, d_fin, d_qc, d_qf):
            b.free()

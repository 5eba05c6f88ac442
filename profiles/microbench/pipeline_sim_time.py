"""Time-weighted simulation of `bench.py --gpus N` (no GPU needed).  pipeline_sim.py steps the executor's readiness rules with
one launch per part per step, which counts the nearly empty launches of pipeline fill and drain like full ones; here every
launch costs what its active tiles cost, the record passes cost what their columns cost, and a part's launch starts when its
predecessor has finished AND the boundary sub-steps it reads have been produced upstream, shipped in batches of
`exchange_rows`, and turned into records.

Cost model, calibrated on one card (profiles/r03_*: 1M reaches, K = 64):
  routing launch: t_task * max(1, active tiles / workgroup slots)         t_task = launch time of a full 4-round launch / 4
  record passes:  (t_in + t_out) per 128 rows, scaled by the part's columns / 1M, spread over the launches of those rows

    python profiles/microbench/pipeline_sim_time.py [reaches per GPU] [t_launch_us t_in_us t_out_us]
    python profiles/microbench/pipeline_sim_time.py --measured profiles/r05_parts_10M.json
        round 5: every part's launch costs are scaled so that the part ALONE takes what it was MEASURED to take on one card
        (`bench.py --sequential-parts 8`: each of the 8 parts of the 10M-reach network routed for the year with its real boundary
        series); what stays simulated is only the overlap between the GPUs -- who waits for whose boundary series, and how long.
"""
import sys
import numpy as np
sys.path.insert(0, '.')
from river_route_amd import synth
from river_route_amd._lib import RR_DEVICE_NONE
from river_route_amd.engine import Plan, partition_forest
from river_route_amd.multi_gpu import split_network

MASK = (1 << 27) - 1
SLOTS = 512


CARD = 288e9


def part_model(spec, T, K, t_task, t_in, t_out):
    with Plan(spec.indptr, spec.indices, device=RR_DEVICE_NONE) as plan:
        ti, L = plan.tile_info(), plan.tile_layout()
        depth = plan.depth
    if K == 0:      # the engine's rule (rr_exec.hpp choose_schedule): 128 ticks per task while the ring stays under a fifth of the card
        ring = lambda k: ((depth - 1 + ti['levels'] * k) // 16 + 32) * 16 * ti['positions'] * 8
        K = 128 if ring(128) <= CARD / 5 and not spec.downstream_parts else 64      # a part that feeds another GPU keeps to 64
        t_task = t_task * (K / 64.0) * (0.865 if K == 128 else 1.0)      # measured: 432 us per launch of 128 ticks against 250 of 64
    lag = L['lag'] & MASK
    tile_of = np.repeat(np.arange(ti['tiles']), np.diff(L['tile_ptr']))
    lo = np.full(ti['tiles'], 1 << 30)
    hi = np.zeros(ti['tiles'], dtype=np.int64)
    np.minimum.at(lo, tile_of, lag)
    np.maximum.at(hi, tile_of, lag)
    level = L['tile_level'].astype(np.int64)
    n_macro = -(-(T + depth - 1) // K)
    n_diags = n_macro + ti['levels'] - 1
    first = level + lo // K
    last = np.minimum(level + n_macro - 1, level + (hi + T - 1) // K)
    active = np.zeros(n_diags + 1)
    np.add.at(active, first, 1)
    np.add.at(active, last + 1, -1)
    active = np.cumsum(active)[:n_diags]
    cost = t_task * np.maximum(1.0, active / SLOTS) * (active > 0)
    cols = spec.n_local / 1e6
    cost += (t_in + t_out) * cols * K / 128.0 * np.clip(active / max(active.max(), 1), 0, 1)      # the passes run while rows flow: not in the empty ends
    real = (L['lag'] & (1 << 28)) == 0
    inv = np.empty(spec.n_local, np.int64)
    inv[L['perm'][real]] = np.flatnonzero(real)
    export_local = spec.n_ghost + np.searchsorted(spec.real_global, spec.export_global)
    pos = inv[export_local]
    skew = int(np.max(level[tile_of[pos]] * K + lag[pos])) if pos.size else 0
    gpos = inv[np.arange(spec.n_ghost)]
    slack = int(np.min(level[tile_of[gpos]] * K + lag[gpos])) if gpos.size else 0
    return dict(cost=cost, n_diags=n_diags, skew=skew, slack=slack, depth=depth, levels=ti['levels'], tiles=ti['tiles'], K=K,
                n=int(spec.real_global.size), ghosts=spec.n_ghost, ups=[s for s, _ in spec.upstream_parts])


def simulate(n, parts, T=35040, K=0, exchange=128, t_launch=234.0, t_in=415.0, t_out=435.0, hop_us=30.0, measured=None):
    net = synth.synth_network(n, order='random')
    has = net.down_index >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
    indices = net.down_index[has].astype(np.int32)
    part_of, _ = partition_forest(indptr, indices, parts)
    t_task = t_launch / 4.0
    P = [part_model(split_network(net.down_index, part_of, p, parts), T, K, t_task, t_in, t_out) for p in range(parts)]
    if measured is not None:      # the part alone takes its measured time; the shape of its launches (fill, drain) is the model's
        for m, rec in zip(P, measured):
            assert rec['reaches'] == m['n'] and rec['ticks_per_launch'] == m['K'], (rec, m['n'], m['K'])
            m['cost'] = m['cost'] * (1e3 * sum(rec['ms_per_year']) / len(rec['ms_per_year']) / m['cost'].sum())
    finish = [None] * parts
    for p, m in enumerate(P):      # parts are numbered upstream-first
        done = np.zeros(m['n_diags'])
        t = 0.0
        for d in range(m['n_diags']):
            start = t
            if m['ups']:
                # launch d reads the boundary sub-steps below (d + 1) K - slack; they arrive in whole exchange batches and
                # become records in whole 128-row batches, 15 rows of which belong to the next one
                need = min(T, max(0, (d + 1) * m['K'] - m['slack']))
                if need > 0:
                    need = min(T, -(-(need + 15) // 128) * 128)
                    need = min(T, -(-need // exchange) * exchange)
                    for q in m['ups']:
                        dq = min(P[q]['n_diags'] - 1, -(-(need + P[q]['skew']) // P[q]['K']))      # upstream launch after which those sub-steps are final
                        start = max(start, finish[q][dq] + hop_us)
            t = start + m['cost'][d]
            done[d] = t
        finish[p] = done
    alone = [float(m['cost'].sum()) for m in P]
    ends = [float(f[-1]) for f in finish]
    print(f'{n} reaches, {parts} parts, T={T}, K={K}, launch {t_launch} us, passes {t_in} + {t_out} us per 128 rows x 1M columns:')
    for p, m in enumerate(P):
        print(f"  part {p}: {m['n']} reaches, {m['ghosts']} boundary inflows from {m['ups']}, depth {m['depth']}, {m['tiles']} tiles / {m['levels']} levels, K = {m['K']}, "
              f"export skew {m['skew']} / ghost slack {m['slack']} ticks, {m['n_diags']} launches: alone {alone[p] / 1e3:.1f} ms, in the run {ends[p] / 1e3:.1f} ms")
    print(f'  last / first finish {max(ends) / min(ends):.3f}; whole job {n * T / (max(ends) * 1e-6):.3e} reach-steps/s = '
          f'{n * T / (max(ends) * 1e-6) / (n / parts * T / (alone[0] * 1e-6)):.2f} x one leaf part alone')


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--measured':
        import json
        line = json.load(open(sys.argv[2]))
        cfg = line['config']
        print(f"part times measured on one card ({sys.argv[2]}: {cfg['parts']} parts of the {cfg['reaches']}-reach network, one after another):")
        simulate(cfg['reaches'], cfg['parts'], T=cfg['runoff_steps'], measured=cfg['per_part'])
        sys.exit(0)
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
    kw = {}
    if len(sys.argv) > 4:
        kw = dict(t_launch=float(sys.argv[2]), t_in=float(sys.argv[3]), t_out=float(sys.argv[4]))
    for parts in (2, 4, 8):
        simulate(per * parts, parts, **kw)

"""Lock-step simulation of the multi-GPU pipeline (no GPU needed): the bench network is partitioned as `bench.py --gpus N`
does, every part gets a host-only plan, and the executor's readiness rules (river_route_amd/csrc/rr_exec.hpp:
session_advance_tile -- a batch of 128 tick-rows becomes records once the lateral rows AND the boundary sub-steps behind it
have arrived; launch d runs once the ticks below (d + 1) K are records and the boundary sub-steps below (d + 1) K - slack are; an export reach in a tile of level l at lag L is
final (d - l) K - L sub-steps into the schedule; finished sub-steps are shipped in batches of `exchange_rows`) are stepped
with one launch per part per step (parts are the same size, so launches take the same time).  Prints, per part, the
launches it needs alone and the step at which it finishes: the ratio is the pipeline's fill cost on top of a perfectly
parallel run.

    python profiles/microbench/pipeline_sim.py [reaches per GPU] > profiles/r02_pipeline_sim.txt
"""
import sys
import numpy as np
sys.path.insert(0, '.')
from river_route_amd import synth
from river_route_amd._lib import RR_DEVICE_NONE
from river_route_amd.engine import Plan, partition_forest
from river_route_amd.multi_gpu import split_network

MASK = (1 << 27) - 1


def simulate(n, parts, T=35040, K=64, batch=128, exchange=128):
    net = synth.synth_network(n, order='random')
    has = net.down_index >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
    indices = net.down_index[has].astype(np.int32)
    part_of, sizes = partition_forest(indptr, indices, parts)
    info = []
    for p in range(parts):
        spec = split_network(net.down_index, part_of, p, parts)
        with Plan(spec.indptr, spec.indices, device=RR_DEVICE_NONE) as plan:
            ti, L = plan.tile_info(), plan.tile_layout()
            depth = plan.depth
        real = (L['lag'] & (1 << 28)) == 0
        inv = np.empty(spec.n_local, np.int64)
        inv[L['perm'][real]] = np.flatnonzero(real)
        tile_of = np.repeat(np.arange(ti['tiles']), np.diff(L['tile_ptr']))
        export_local = spec.n_ghost + np.searchsorted(spec.real_global, spec.export_global)
        pos = inv[export_local]
        skew = int(np.max(L['tile_level'][tile_of[pos]] * K + (L['lag'][pos] & MASK))) if pos.size else 0
        gpos = inv[np.arange(spec.n_ghost)]
        slack = int(np.min(L['tile_level'][tile_of[gpos]] * K + (L['lag'][gpos] & MASK))) if gpos.size else 0
        n_macro = -(-(T + depth - 1) // K)
        info.append(dict(p=p, n=int(spec.real_global.size), ghosts=spec.n_ghost, exports=int(pos.size), depth=depth, levels=ti['levels'],
                         tiles=ti['tiles'], skew=skew, slack=slack, n_diags=n_macro + ti['levels'] - 1, ups=[s for s, _ in spec.upstream_parts]))
    sent = [0] * parts        # sub-steps of each part's export series shipped downstream
    diag = [0] * parts
    done_step = [None] * parts
    step = 0
    while any(d is None for d in done_step):
        step += 1
        new_sent = list(sent)
        for i in info:
            p = i['p']
            if done_step[p] is not None:
                continue
            ghost_ready = min([sent[s] for s in i['ups']], default=T)
            # boundary sub-steps that are records: whole batches of the ghost series (lateral rows are all resident); a
            # ghost is first read `slack` ticks into the schedule
            batches = T // batch + 1 if ghost_ready >= T else ghost_ready // batch
            have = T if batches * batch >= T else max(0, batches * batch - 15)
            if have >= min(max(0, (diag[p] + 1) * K - i['slack']), T):
                diag[p] += 1
            if diag[p] >= i['n_diags']:
                done_step[p] = step
                ready = T
            else:
                ready = max(0, min(T, diag[p] * K - i['skew']))
            new_sent[p] = T if ready >= T else (ready // exchange) * exchange
        sent = new_sent
    print(f'{n} reaches, {parts} parts, T={T}, K={K}:')
    for i in info:
        print(f"  part {i['p']}: {i['n']} reaches, {i['ghosts']} boundary inflows from parts {i['ups']}, {i['exports']} exports, depth {i['depth']}, "
              f"{i['tiles']} tiles in {i['levels']} levels, export skew {i['skew']} / ghost slack {i['slack']} ticks: {i['n_diags']} launches alone, "
              f"finishes at step {done_step[i['p']]} ({done_step[i['p']] / i['n_diags']:.3f}x)")
    first, last = min(done_step), max(done_step)
    print(f'  last part finishes {last / first:.3f}x after the first; the slowest part alone needs {max(i["n_diags"] for i in info)} launches')


if __name__ == '__main__':
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
    for parts in (2, 4, 8):
        simulate(per * parts, parts)

"""Lock-step simulation of the multi-GPU pipeline (no GPU needed): the bench network is partitioned as `bench.py --gpus N`
does, every part gets a host-only plan, and the executor's readiness rules (river_route_amd/csrc/rr_engine.hip:
session_advance_wave -- a diagonal of the time-tiled schedule may launch once the boundary sub-steps it reads have
arrived; exports become final `wave_export_skew` ticks behind the schedule; batches of 128 sub-steps are shipped) are
stepped with one launch per part per step.  Prints, per part, the launches it needs alone and the step at which it
finishes: the ratio is the pipeline's fill cost on top of a perfectly parallel run."""
import numpy as np, sys, time
sys.path.insert(0, '.')
from river_route_amd import synth
from river_route_amd.engine import partition_forest, Plan
from river_route_amd.multi_gpu import split_network
from river_route_amd._lib import RR_DEVICE_NONE

def simulate(n, parts, T=35040, K=16, BS=2048, batch=128):
    net = synth.synth_network(n, order='random')
    has = net.down_index >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
    indices = net.down_index[has].astype(np.int32)
    part_of, sizes = partition_forest(indptr, indices, parts)
    info = []
    for p in range(parts):
        spec = split_network(net.down_index, part_of, p, parts)
        plan = Plan(spec.indptr, spec.indices, device=RR_DEVICE_NONE)
        perm, lag, child_ptr = plan.layout()
        inv = np.empty_like(perm); inv[perm] = np.arange(perm.size)
        n_loc = perm.size
        ng = spec.n_ghost
        nb = (n_loc + BS - 1) // BS
        ghost_pos = inv[np.arange(ng)]
        export_local = ng + np.searchsorted(spec.real_global, spec.export_global)
        exp_pos = inv[export_local]
        slack = int(np.min(ghost_pos // BS * K + lag[ghost_pos])) if ng else 0
        # per upstream part slack (min over its ghosts)
        ups = {}
        for src, cols in spec.upstream_parts:
            gp = ghost_pos[cols]
            ups[src] = int(np.min(gp // BS * K + lag[gp]))
        skew = int(np.max(exp_pos // BS * K + lag[exp_pos])) if export_local.size else 0
        depth = int(lag.max()) + 1
        n_chunks = (T + depth - 1 + K - 1) // K
        info.append(dict(nb=nb, slack=slack, ups=ups, skew=skew, depth=depth, n_diags=n_chunks + nb - 1, n=n_loc))
        plan.close()
    # lockstep simulation in launch units
    d = [0] * parts
    sent = [0] * parts      # export sub-steps shipped (batched)
    t = 0
    done = [False] * parts
    finish = [0] * parts
    while not all(done) and t < 100000:
        t += 1
        export_ready = [T if d[p] >= info[p]['n_diags'] else max(0, d[p] * K - info[p]['skew']) for p in range(parts)]
        for p in range(parts):
            r = min(export_ready[p], T)
            while r - sent[p] >= batch or (r >= T and sent[p] < T):
                sent[p] = min(sent[p] + batch, r, T)
        nd = list(d)
        for p in range(parts):
            if done[p]: continue
            ok = True
            for q, sl in info[p]['ups'].items():
                need = min((d[p] + 1) * K - info[p]['slack'], T)     # engine uses the part-wide minimum slack
                if sent[q] < need: ok = False
            if ok:
                nd[p] = d[p] + 1
                if nd[p] >= info[p]['n_diags']:
                    done[p] = True; finish[p] = t
        d = nd
    base = max(i['n_diags'] for i in info)
    return info, finish, base

for n, parts in ((2_000_000, 2), (4_000_000, 4), (8_000_000, 8)):
    t0 = time.time()
    info, finish, base = simulate(n, parts)
    print(n, parts, 'launches alone', [i['n_diags'] for i in info], 'finish', finish, 'ratio %.2f' % (max(finish) / base),
          'slack', [i['slack'] for i in info], 'skew', [i['skew'] for i in info], 'depth', [i['depth'] for i in info], f'{time.time()-t0:.0f}s')

"""One part of the 8-GPU bench run on one GPU: the LAST part (main stems, deepest) and the FIRST part of the 10M-reach (BASELINE config 5)
network cut into 8, each routed alone for one year with all boundary inflow already present (zeros).  Shows what each
of the 8 GPUs has to do per pass, which routing kernel its part gets (RR_VERBOSE line) and how long it takes."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from river_route_amd import synth
from river_route_amd.engine import partition_forest
from river_route_amd.multi_gpu import split_network, HipPartEngine

world, per = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
n, T, nsub, dt = per * world, 35040, 1, 900.0
net = synth.synth_network(n, order='random')
has = net.down_index >= 0
indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
indices = net.down_index[has].astype(np.int32)
part_of, sizes = partition_forest(indptr, indices, world)
r = dt / net.k
den = r + 2.0 * (1.0 - net.x)
c1, c2, c3 = (r - 2.0 * net.x) / den, (r + 2.0 * net.x) / den, (2.0 * (1.0 - net.x) - r) / den
for part in (world - 1, 0):
    spec = split_network(net.down_index, part_of, part, world)
    idx = (np.arange(96, dtype=np.uint64)[:, None] * np.uint64(n)) + spec.real_global.astype(np.uint64)[None, :]
    lateral = dt * synth.u01(synth.FORCING_SEED, idx)
    eng = HipPartEngine(spec, c1, c2, c3, (c1 + c2) / dt, np.zeros(n), lateral, T, nsub, 0)
    ts = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.begin(); eng.advance(T, T * nsub); eng.end()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f'part {part}: {spec.real_global.size} reaches + {spec.n_ghost} ghosts, depth {eng.plan.depth}, exports {spec.export_global.size}: '
          + ' '.join(f'{t * 1e3:.1f}' for t in ts) + ' ms per one-year pass', flush=True)
    del eng
    torch.cuda.empty_cache()

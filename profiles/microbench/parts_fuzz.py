"""Randomised parity sweep of the partitioned path against the oracle on the UNDIVIDED network (round 5; a development aid): random networks cut into 2-8 parts by
rr_partition_forest, every part through rr_stream_begin / advance / end with its boundary series exchanged in batches of random length (multi_gpu.run_in_process: the
distributed driver without the network), post-order (direct row path with boundary reaches) or random order (records), random task lengths.
usage: python profiles/microbench/parts_fuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle
from river_route_amd import synth
from river_route_amd.engine import partition_forest
from river_route_amd.multi_gpu import HipPartEngine, run_in_process, split_network

def csc(down):
    has = down >= 0
    return np.concatenate([[0], np.cumsum(has)]).astype(np.int32), down[has].astype(np.int32)

cases, seed0 = (int(sys.argv[1]) if len(sys.argv) > 1 else 30), (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rng = np.random.default_rng(seed0)
for case in range(cases):
    n = int(rng.choice([3000, 20_000, 60_000, 150_000]))
    order = str(rng.choice(['postorder', 'postorder', 'random']))
    chainy = bool(rng.integers(0, 4) == 0)
    seed = int(rng.integers(1, 1 << 20))
    net = (synth.synth_network_chain(n, p_chain=0.3, n_outlets=int(rng.integers(1, 5)), p_third=0.04, seed=seed, order=order) if chainy else synth.synth_network(n, seed=seed, order=order))
    parts = int(rng.integers(2, 9))
    T = int(rng.choice([40, 96, 200, 700, 1300]))
    chunk = int(rng.choice([8, 32, 64, 128]))
    K = str(rng.choice(['', '', '64', '256', '1024']))
    if K: os.environ['RR_WAVE_K'] = K
    else: os.environ.pop('RR_WAVE_K', None)
    print(f'case {case:3d}: n={n} {order} chainy={chainy} seed={seed} parts={parts} T={T} chunk={chunk} K={K or "-"} ...', flush=True)
    indptr, indices = csc(net.down_index)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0)
    c4 = (c1 + c2) / 900.0
    q0 = 4.0 * synth.u01(8, np.arange(n))
    ql = synth.synth_qlateral(n, 0, T)
    q_ref, d_ref = q0.copy(), np.zeros((T, n))
    oracle.rapid_route(indptr, indices, -c1[indices], c2, c3, c4, q_ref, ql, d_ref, 1)
    part_of, _ = partition_forest(indptr, indices, parts)
    specs = [split_network(net.down_index, part_of, p, parts) for p in range(parts)]
    engines = [HipPartEngine(s, c1, c2, c3, c4, q0, ql[:, s.real_global], T, 1, 0, out_rows=T) for s in specs]
    run_in_process(engines, specs, T, 1, chunk)
    kernels = [e.plan.last_kernel() for e in engines]
    q, d = np.zeros(n), np.zeros((T, n))
    for s, e in zip(specs, engines):
        q[s.real_global] = e.final_state()
        d[:, s.real_global] = e.discharge.cpu().numpy()[:, s.n_ghost:]
        e.close()
    scale = float(np.abs(d_ref).max())
    err = float(np.abs(d - d_ref).max()) / scale
    assert np.allclose(d, d_ref, rtol=1e-10, atol=1e-10 * scale) and np.allclose(q, q_ref, rtol=1e-10, atol=1e-10 * scale), f'case {case}: max diff {err:.3e}'
    print(f'case {case:3d}: n={n:6d} {order:9s} {"chain" if chainy else "remy ":5s} parts={parts} ghosts={[s.n_ghost for s in specs]} T={T:4d} chunk={chunk:3d} K={K or "-":>4s} kernels={"".join(k[0] for k in kernels)} max diff {err:.1e}', flush=True)
print('all cases agree with the oracle on the undivided network')

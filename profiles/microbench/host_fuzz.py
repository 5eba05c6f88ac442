"""Randomised parity sweep of the reference-shaped kernel functions with numpy arrays (round 5; a development aid): kernels.rapid_route / muskingum_route / unit_route --
host pointers through the C ABI, the PCIe pipeline or the streaming kernel -- on random networks in any params order, 1-4 sub-steps, 1 ... 700 rows, two consecutive
calls on the same arrays (state carried in place, as the reference's loop over files does).  usage: python profiles/microbench/host_fuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np
from oracle import oracle
from river_route_amd import kernels, synth
from tests_support import unit_split_arrays

def close(a, b, what):
    scale = max(1e-300, float(np.abs(b).max()))
    err = float(np.abs(a - b).max()) / scale
    assert np.allclose(a, b, rtol=1e-10, atol=1e-10 * scale), f'{what}: max diff {err:.3e} of the largest value'
    return err

cases, seed0 = (int(sys.argv[1]) if len(sys.argv) > 1 else 40), (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rng = np.random.default_rng(seed0)
for case in range(cases):
    n = int(rng.choice([1, 2, 9, 300, 3000, 40000]))
    order = str(rng.choice(['random', 'postorder', 'levels', 'bfs']))
    seed = int(rng.integers(1, 1 << 20))
    net = synth.synth_network(n, seed=seed, order=order)
    has = net.down_index >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32); indices = net.down_index[has].astype(np.int32)
    nsub = int(rng.choice([1, 1, 2, 4]))
    mode = str(rng.choice(['rapid', 'muskingum', 'unit']))
    Ts = (int(rng.choice([1, 7, 40, 130, 700])), int(rng.choice([1, 33, 96])))
    print(f'case {case:3d}: n={n} {order} seed={seed} {mode} nsub={nsub} T={Ts} ...', flush=True)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0 / nsub)
    q0 = 2.0 * synth.u01(3, np.arange(n))
    worst = 0.0
    if mode == 'unit':
        hw_idx, inner_idx, A_in, A_hw = unit_split_arrays(indptr, indices, n)
        c1i, c2i, c3i = c1[inner_idx], c2[inner_idx], c3[inner_idx]
        args = (A_in.indptr, A_in.indices, -c1i[A_in.indices], A_in.indptr, A_in.indices, A_in.data, A_hw.indptr, A_hw.indices, A_hw.data, c1i, c2i, c3i, hw_idx, inner_idx)
        qc, qf = q0[inner_idx].copy(), q0[inner_idx].copy()
        qc_ref, qf_ref = qc.copy(), qf.copy()
        t0 = 0
        for T in Ts:
            conv = np.abs(synth.synth_qlateral(n, t0, t0 + T)); t0 += T
            d, d_ref = np.zeros((T, n)), np.zeros((T, n))
            oracle.unit_route(*args, qc_ref, qf_ref, conv, d_ref, nsub)
            kernels.unit_route(*args, qc, qf, conv, d, nsub)
            worst = max(worst, close(d, d_ref, f'case {case} unit discharge'))
            if inner_idx.size: close(qf, qf_ref, f'case {case} q_full'); close(qc, qc_ref, f'case {case} q_ch')
    else:
        c4 = (c1 + c2) / 900.0
        q, q_ref = q0.copy(), q0.copy()
        t0 = 0
        for T in Ts:
            d, d_ref = np.zeros((T, n)), np.zeros((T, n))
            if mode == 'rapid':
                ql = synth.synth_qlateral(n, t0, t0 + T); t0 += T
                oracle.rapid_route(indptr, indices, -c1[indices], c2, c3, c4, q_ref, ql, d_ref, nsub)
                kernels.rapid_route(indptr, indices, -c1[indices], c2, c3, c4, q, ql, d, nsub)
            else:
                oracle.muskingum_route(indptr, indices, -c1[indices], c2, c3, q_ref, d_ref, T, nsub)
                kernels.muskingum_route(indptr, indices, -c1[indices], c2, c3, q, d, T, nsub)
            worst = max(worst, close(d, d_ref, f'case {case} {mode} discharge'))
            close(q, q_ref, f'case {case} state')
    kernels.clear_plan_cache()
    print(f'          max diff {worst:.1e}', flush=True)
print('all cases agree with the oracle')

"""Randomised parity sweep of the direct row path against the oracle (round 5; a development aid, not a test: 60 small post-order networks of random size,
shape and seed -- Remy trees and chain-grown forests with three-way confluences --, RapidMuskingum / channel-only / UnitMuskingum, 1-4 sub-steps, float64 or float32
rows, random task lengths; every row and the final state at rtol 1e-10).  usage: python profiles/microbench/direct_fuzz.py [cases] [seed] [params order: postorder (default) | random | levels | bfs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np
from oracle import oracle
from river_route_amd import synth
from river_route_amd.engine import DeviceBuffer, Plan
from tests_support import unit_split_arrays

def csc(down):
    has = down >= 0
    return np.concatenate([[0], np.cumsum(has)]).astype(np.int32), down[has].astype(np.int32)

def close(a, b, what):
    scale = max(1e-300, float(np.abs(b).max()))
    err = float(np.abs(a - b).max()) / scale
    assert np.allclose(a, b, rtol=1e-10, atol=1e-10 * scale), f'{what}: max diff {err:.3e} of the largest value'
    return err

cases, seed0 = (int(sys.argv[1]) if len(sys.argv) > 1 else 60), (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
order = sys.argv[3] if len(sys.argv) > 3 else 'postorder'      # 'random' / 'levels' / 'bfs': the same sweep through records (k_tile, k_tick)
rng = np.random.default_rng(seed0)
kinds = {}
for case in range(cases):
    n = int(rng.choice([1, 2, 3, 7, 40, 257, 300, 700, 2000, 6000, 20000]))
    chainy = bool(rng.integers(0, 3) == 0) and n >= 40
    seed = int(rng.integers(1, 1 << 20))
    net = (synth.synth_network_chain(n, p_chain=float(rng.choice([0.2, 0.5, 0.8])), n_outlets=int(rng.integers(1, 6)), p_third=0.05, seed=seed, order=order) if chainy
           else synth.synth_network(n, seed=seed, order=order))
    indptr, indices = csc(net.down_index)
    nsub = int(rng.choice([1, 1, 2, 3, 4]))
    mode = str(rng.choice(['rapid', 'rapid', 'muskingum', 'unit']))
    T = int(rng.choice([8, 33, 100, 257, 600]))
    K = str(rng.choice(['', '32', '64', '256', '1024']))
    if K: os.environ['RR_WAVE_K'] = K
    else: os.environ.pop('RR_WAVE_K', None)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0 / nsub)
    q0 = 2.0 * synth.u01(3, np.arange(n))
    with Plan(indptr, indices) as plan:
        info = plan.direct_info()
        d_out = DeviceBuffer(T * n * 8)
        if mode == 'unit':
            plan.set_coeffs(-c1[indices], c2, c3, None)
            hw_idx, inner_idx, A_in, A_hw = unit_split_arrays(indptr, indices, n)
            ni = inner_idx.size
            c1i, c2i, c3i = c1[inner_idx], c2[inner_idx], c3[inner_idx]
            conv = oracle.UnitHydrograph(synth.synth_uh_kernel(n, 5)).convolve(synth.synth_runoff_depth(n, 0, T))
            qc, qf, d_ref = q0[inner_idx].copy(), q0[inner_idx].copy(), np.zeros((T, n))
            oracle.unit_route(A_in.indptr, A_in.indices, -c1i[A_in.indices], A_in.indptr, A_in.indices, A_in.data, A_hw.indptr, A_hw.indices, A_hw.data,
                              c1i, c2i, c3i, hw_idx, inner_idx, qc, qf, conv, d_ref, nsub)
            d_conv, d_qc, d_qf = DeviceBuffer(T * n * 8).upload(conv), DeviceBuffer(max(ni, 1) * 8), DeviceBuffer(max(ni, 1) * 8)
            d_qc.upload(q0[inner_idx].copy()); d_qf.upload(q0[inner_idx].copy())
            plan.unit_route_dev(d_qc, d_qf, d_conv, T, d_out, T, T, nsub)
            e = close(d_out.download(np.float64, (T, n)), d_ref, f'case {case} unit discharge')
            if ni:
                close(d_qf.download(np.float64, (ni,)), qf, f'case {case} q_full'); close(d_qc.download(np.float64, (ni,)), qc, f'case {case} q_ch')
            for b in (d_conv, d_qc, d_qf): b.free()
        else:
            c4 = (c1 + c2) / 900.0
            plan.set_coeffs(-c1[indices], c2, c3, c4 if mode == 'rapid' else None)
            q_ref, d_ref = q0.copy(), np.zeros((T, n))
            d_q = DeviceBuffer(n * 8).upload(q0)
            if mode == 'rapid':
                ql = synth.synth_qlateral(n, 0, T)
                in32 = bool(rng.integers(0, 2))
                if in32: ql = ql.astype(np.float32)
                oracle.rapid_route(indptr, indices, -c1[indices], c2, c3, c4, q_ref, ql.astype(np.float64), d_ref, nsub)
                d_ql = DeviceBuffer(ql.nbytes).upload(ql)
                try:
                    if in32: plan.rapid_route_f32in_dev(d_q, d_ql, T, T, nsub, discharge=d_out, out_rows=T)
                    else: plan.rapid_route_dev(d_q, d_ql, T, d_out, T, T, nsub)
                except Exception as exc:      # the float32 form is refused where the call is not time-tiled (a few sub-steps on records): the float64 form then, as the routers do
                    from river_route_amd._lib import RR_E_UNSUPPORTED
                    if not in32 or getattr(exc, 'code', None) != RR_E_UNSUPPORTED: raise
                    d_ql.free(); d_ql = DeviceBuffer(T * n * 8).upload(ql.astype(np.float64))
                    plan.rapid_route_dev(d_q, d_ql, T, d_out, T, T, nsub)
                d_ql.free()
            else:
                oracle.muskingum_route(indptr, indices, -c1[indices], c2, c3, q_ref, d_ref, T, nsub)
                plan.muskingum_route_dev(d_q, d_out, T, T, nsub)
            e = close(d_out.download(np.float64, (T, n)), d_ref, f'case {case} {mode} discharge')
            close(d_q.download(np.float64, (n,)), q_ref, f'case {case} state')
            d_q.free()
        kern = plan.last_kernel()
        kinds[kern] = kinds.get(kern, 0) + 1
        d_out.free()
    print(f'case {case:3d}: n={n:6d} {"chain" if chainy else "remy ":5s} {mode:9s} nsub={nsub} T={T:4d} K={K or "-":>4s} direct_ok={info["ok"]!s:5s} ran {kern:6s} max diff {e:.1e}', flush=True)
print('all cases agree with the oracle;', kinds)

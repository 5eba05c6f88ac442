"""Randomised parity sweep of UnitMuskingum with its convolution (round 5; a development aid): rr_unit_route_uh_dev / rr_unit_route_uh_f32in_dev on random networks in post-order
(the convolution pass + the direct row path) or in a random order (the convolution fused into the record in-pass), random kernel lengths (1 ... 60 taps), 1-3 sub-steps,
two consecutive files of random lengths (also shorter than the kernel): discharge, router state, channel state and the convolution's carry-over against the oracle.
usage: python profiles/microbench/unit_fuzz.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import numpy as np
from oracle import oracle
from river_route_amd import synth
from river_route_amd._lib import RR_E_UNSUPPORTED
from river_route_amd.engine import DeviceBuffer, Plan
from tests_support import unit_split_arrays

def close(a, b, what):
    scale = max(1e-300, float(np.abs(b).max()))
    err = float(np.abs(a - b).max()) / scale
    assert np.allclose(a, b, rtol=1e-10, atol=1e-10 * scale), f'{what}: max diff {err:.3e} of the largest value'
    return err

cases, seed0 = (int(sys.argv[1]) if len(sys.argv) > 1 else 40), (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rng = np.random.default_rng(seed0)
kinds = {}
for case in range(cases):
    n = int(rng.choice([40, 300, 2000, 6000, 30000]))
    order = str(rng.choice(['postorder', 'postorder', 'random']))
    seed = int(rng.integers(1, 1 << 20))
    net = synth.synth_network(n, seed=seed, order=order)
    has = net.down_index >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32); indices = net.down_index[has].astype(np.int32)
    nsub = int(rng.choice([1, 1, 2, 3]))
    n_ks = int(rng.choice([1, 5, 12, 33, 48, 60]))
    in32 = bool(rng.integers(0, 2))
    Ts = (int(rng.choice([40, 100, 257, 600])), int(rng.choice([8, 20, 64])))
    K = str(rng.choice(['', '', '64', '256']))
    if K: os.environ['RR_WAVE_K'] = K
    else: os.environ.pop('RR_WAVE_K', None)
    print(f'case {case:3d}: n={n} {order} seed={seed} nsub={nsub} n_ks={n_ks} in32={in32} T={Ts} K={K or "-"} ...', flush=True)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0 / nsub)
    hw_idx, inner_idx, A_in, A_hw = unit_split_arrays(indptr, indices, n)
    c1i, c2i, c3i = c1[inner_idx], c2[inner_idx], c3[inner_idx]
    args = (A_in.indptr, A_in.indices, -c1i[A_in.indices], A_in.indptr, A_in.indices, A_in.data, A_hw.indptr, A_hw.indices, A_hw.data, c1i, c2i, c3i, hw_idx, inner_idx)
    kern = synth.synth_uh_kernel(n, n_ks)
    uh = oracle.UnitHydrograph(kern)
    state_ref = 3.0 * synth.u01(7, np.arange(n))
    ni = inner_idx.size
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(-c1[indices], c2, c3, None)
        Tm = max(Ts)
        d_kern, d_state = DeviceBuffer(kern.nbytes).upload(kern), DeviceBuffer(kern.nbytes).upload(np.zeros_like(kern))
        d_depth, d_out, d_fin = DeviceBuffer(Tm * n * 8), DeviceBuffer(Tm * n * 8), DeviceBuffer(n * 8)
        d_qc, d_qf = DeviceBuffer(max(ni, 1) * 8), DeviceBuffer(max(ni, 1) * 8)
        state = state_ref.copy()
        t0 = 0
        ran = []
        for f, Tf in enumerate(Ts):
            depth = synth.synth_runoff_depth(n, t0, t0 + Tf); t0 += Tf
            if in32: depth = depth.astype(np.float32)
            conv_ref = uh.convolve(depth.astype(np.float64))
            qc_ref, qf_ref, d_ref = state_ref[inner_idx].copy(), state_ref[inner_idx].copy(), np.zeros((Tf, n))
            oracle.unit_route(*args, qc_ref, qf_ref, conv_ref, d_ref, nsub)
            state_ref[hw_idx], state_ref[inner_idx] = conv_ref[-1][hw_idx], qf_ref
            d_depth.upload(depth)
            d_qc.upload(state[inner_idx].copy()); d_qf.upload(state[inner_idx].copy())
            call = plan.unit_route_uh_f32in_dev if in32 else plan.unit_route_uh_dev
            try:
                call(d_qc, d_qf, d_fin, d_kern, d_state, n_ks, d_depth, Tf, nsub, discharge=d_out)
                ran.append(plan.last_kernel())
                state = d_fin.download(np.float64, (n,))
                e = close(d_out.download(np.float64, (Tf, n)), d_ref, f'case {case} file {f} discharge')
                close(state, state_ref, f'case {case} file {f} router state')
                if ni: close(d_qc.download(np.float64, (ni,)), qc_ref, f'case {case} file {f} q_ch')
                close(d_state.download(np.float64, kern.shape), uh.state, f'case {case} file {f} UH state')
            except Exception as exc:      # a call too short for the time-tiled kernel is refused (the two-call form then): stop this case there
                if getattr(exc, 'code', None) != RR_E_UNSUPPORTED: raise
                ran.append('refused'); break
        for k in ran: kinds[k] = kinds.get(k, 0) + 1
        for b in (d_kern, d_state, d_depth, d_out, d_fin, d_qc, d_qf): b.free()
    print(f'          ran {ran}', flush=True)
print('all cases agree with the oracle;', kinds)

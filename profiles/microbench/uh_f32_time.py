"""BASELINE config 4's call (UnitMuskingum, 1M reaches, 48-step kernel, 3,504 rows) from float64 and from float32 runoff depths,
float64 rows out and hourly float32 means out: rr_unit_route_uh_dev against rr_unit_route_uh_f32in_dev.
    python profiles/microbench/uh_f32_time.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from river_route_amd import synth
from river_route_amd.engine import Plan

n, T, n_ks = 1_000_000, 3504, 48
net = synth.synth_network(n)
has = net.down_index >= 0
indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32); indices = net.down_index[has].astype(np.int32)
r = 900.0 / net.k; den = r + 2 * (1 - net.x)
c1, c2, c3 = (r - 2 * net.x) / den, (r + 2 * net.x) / den, (2 * (1 - net.x) - r) / den
dev = torch.device('cuda:0')
kern = torch.from_numpy(synth.synth_uh_kernel(n, n_ks)).to(dev)
depth64 = torch.rand((T, n), dtype=torch.float64, device=dev) * 1e-3
depth32 = depth64.to(torch.float32)
with Plan(indptr, indices) as plan:
    plan.set_coeffs(-c1[indices], c2, c3, None)
    ni = plan.n_inner
    st = torch.cuda.current_stream().cuda_stream
    for name, depth, call in (('float64 depths', depth64, plan.unit_route_uh_dev), ('float32 depths', depth32, plan.unit_route_uh_f32in_dev)):
        for out_name, kw in (('float64 rows out', dict(discharge=torch.empty((T, n), dtype=torch.float64, device=dev))),
                             ('hourly float32 means out', dict(discharge32=torch.empty((T // 4, n), dtype=torch.float32, device=dev), factor=4))):
            state = torch.zeros_like(kern); qc = torch.zeros(ni, dtype=torch.float64, device=dev); qf = torch.zeros_like(qc)
            fin = torch.empty(n, dtype=torch.float64, device=dev)
            best = 1e9
            for rep in range(4):
                state.zero_(); qc.zero_(); qf.zero_()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                call(qc, qf, fin, kern, state, n_ks, depth, T, 1, stream=st, **kw)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            print(f'{name}, {out_name}: {best * 1e3:.2f} ms, {n * T / best:.3e} reach-steps/s', flush=True)

"""Which float32 side slows k_direct down (round 5): the same 8,192-row call at 1M reaches in post-order with float64 / float32 lateral rows in and
float64 rows / float32 means of four out.  usage: python profiles/microbench/direct_f32_variants.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from river_route_amd import synth
from river_route_amd.engine import Plan
from bench import csc_from_down, muskingum_coefficients
n, T, rows, dt = 1_000_000, 8192, 1024, 900.0
dev = torch.device('cuda:0')
net = synth.synth_network(n, order='postorder')
indptr, indices = csc_from_down(net.down_index)
c1, c2, c3 = muskingum_coefficients(net.k, net.x, dt)
plan = Plan(indptr, indices)
plan.set_coeffs(-c1[indices], c2, c3, (c1 + c2) / dt)
plan.set_options(rows_per_chunk=16, sample_every=128)
ql64 = synth.synth_qlateral_torch(n, 0, rows, dev, dt=dt)
ql32 = ql64.to(torch.float32)
out64 = torch.zeros((rows, n), dtype=torch.float64, device=dev)
out32 = torch.zeros((T // 4, n), dtype=torch.float32, device=dev)
q = torch.zeros(n, dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
def run(in32, o32):
    q.zero_()
    if in32: plan.rapid_route_f32in_dev(q, ql32, rows, T, 1, **(dict(discharge32=out32, factor=4) if o32 else dict(discharge=out64, out_rows=rows)), stream=st)
    elif o32: plan.rapid_route_f32_dev(q, ql64, rows, out32, T, 1, factor=4, stream=st)
    else: plan.rapid_route_dev(q, ql64, rows, out64, rows, T, 1, stream=st)
for in32 in (False, True):
    for o32 in (False, True):
        run(in32, o32); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(in32, o32); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
        p = plan.profile(); a = plan.profile_aux()
        print(f'in {"f32" if in32 else "f64"} out {"f32 means of 4" if o32 else "f64"}: {ms:7.2f} ms  kernel {plan.last_kernel()} K={p["ticks_per_launch"]}  k_direct {p["sampled_ms"] / max(1, p["brackets"]) * 1e3:8.1f} us per launch  ' +
              '  '.join(f'{k} {v["sampled_ms"] / max(1, v["sampled"]) * 1e3:.1f} us' for k, v in a.items()))

"""Channel-only Muskingum (muskingum_route, river_route/routers/_numba_kernels.py:8-46) at 1M reaches, one routing step per output
row, a year of 15-minute rows into a 128-row cyclic sink: the time of rr_muskingum_route_dev (no lateral rows, so no in-pass).
    python profiles/microbench/channel_only_time.py            (RR_TILE_LEAN=0: the general tick)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import torch
from river_route_amd import synth
from river_route_amd.engine import Plan

n, T = 1_000_000, 35_040
net = synth.synth_network(n)
has = net.down_index >= 0
indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32); indices = net.down_index[has].astype(np.int32)
r = 900.0 / net.k; den = r + 2 * (1 - net.x)
c1, c2, c3 = (r - 2 * net.x) / den, (r + 2 * net.x) / den, (2 * (1 - net.x) - r) / den
dev = torch.device('cuda:0')
with Plan(indptr, indices) as plan:
    plan.set_coeffs(-c1[indices], c2, c3, None)
    q0 = torch.from_numpy(1.0 + synth.u01(3, np.arange(n))).to(dev)
    q = torch.empty_like(q0)
    out = torch.zeros((128, n), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(4):
        q.copy_(q0); torch.cuda.synchronize(); t0 = time.perf_counter()
        plan.muskingum_route_dev(q, out, 128, T, 1, st)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f'channel-only, 1M reaches x {T} rows: {dt * 1e3:.1f} ms, {n * T / dt:.3e} reach-steps/s, K = {plan.profile()["ticks_per_launch"]}', flush=True)

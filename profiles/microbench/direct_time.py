"""Times rr_rapid_route_dev on a post-order network through the direct row path and, with RR_DIRECT=0, through the record path.
usage: direct_time.py [n] [T] [forcing_rows] [sink_rows] [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from river_route_amd import synth  # noqa: E402
from river_route_amd.engine import Plan  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 35_040
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
sink = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
order = os.environ.get('ORDER', 'postorder')
dev = torch.device('cuda', 0)
net = synth.synth_network(n, order=order)
has = net.down_index >= 0
indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
indices = net.down_index[has].astype(np.int32)
r = 900.0 / net.k
den = r + 2.0 * (1.0 - net.x)
c1, c2, c3 = (r - 2.0 * net.x) / den, (r + 2.0 * net.x) / den, (2.0 * (1.0 - net.x) - r) / den
plan = Plan(indptr, indices)
plan.set_coeffs(-c1[indices], c2, c3, (c1 + c2) / 900.0)
plan.set_options(sample_every=128)
ql = synth.synth_qlateral_torch(n, 0, min(rows, T), dev)
out = torch.zeros((min(sink, T), n), dtype=torch.float64, device=dev)
q = torch.zeros(n, dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
sched = plan.reserve(0, T, 1)
for rep in range(reps + 1):
    q.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.rapid_route_dev(q, ql, ql.shape[0], out, out.shape[0], T, 1, stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rep:
        p = plan.profile()
        print(f'{order} n={n} T={T} kernel={plan.last_kernel()} K={p["ticks_per_launch"]} ring={sched["ring_bytes"] / 1e9:.1f} GB: {dt * 1e3:.1f} ms '
              f'{n * T / dt:.3e} reach-steps/s; sampled launches {p["brackets"]} avg {p["sampled_ms"] / max(1, p["brackets"]) * 1e3:.1f} us; direct {plan.direct_info()}', flush=True)

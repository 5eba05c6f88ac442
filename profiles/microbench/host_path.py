"""Times the host-pointer entry point rr_rapid_route (numpy arrays in/out, what river_route_amd.kernels.rapid_route
calls): the PCIe-inclusive rate noted in DESIGN.md.    python host_path.py [rows] [reaches]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from river_route_amd import synth
from river_route_amd.engine import Plan
T = int(sys.argv[1]) if len(sys.argv) > 1 else 192
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
net = synth.synth_network(n)
has = net.down_index >= 0
indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32); indices = net.down_index[has].astype(np.int32)
r = 900.0 / net.k; den = r + 2 * (1 - net.x)
c1, c2, c3 = (r - 2 * net.x) / den, (r + 2 * net.x) / den, (2 * (1 - net.x) - r) / den
plan = Plan(indptr, indices); plan.set_coeffs(-c1[indices], c2, c3, (c1 + c2) / 900.0)
ql = synth.synth_qlateral(n, 0, T); d = np.zeros((T, n)); q = np.zeros(n)
plan.rapid_route(q, ql, d, 1)
for rep in range(2):
    q[:] = 0; t0 = time.perf_counter(); plan.rapid_route(q, ql, d, 1); dt = time.perf_counter() - t0
    print(f'host path: n={n} T={T}: {dt*1e3:.1f} ms -> {n*T/dt:.3e} reach-steps/s, {2*ql.nbytes/dt/1e9:.1f} GB/s over PCIe')

"""Topology sensitivity: 1M reaches grown with synth_network_chain at increasing p_chain (depth 2k ... 24k),
several outlets, some three-way confluences -- bench-style timing of one year (T = 35,040, 288-row forcing ring, 128-row
sink) with the schedule the engine chose (kernel, K, ring) next to the Remy tree of the headline.
    python profiles/microbench/depth_sweep.py"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import torch
from river_route_amd import synth
from river_route_amd.engine import Plan, MODE_RAPID

n, T, dt = 1_000_000, 35_040, 900.0
dev = torch.device('cuda', 0)
ql = synth.synth_qlateral_torch(n, 0, 288, dev)
out = torch.zeros((128, n), dtype=torch.float64, device=dev)
q_t = torch.zeros(n, dtype=torch.float64, device=dev)
cases = [('remy tree (headline)', None)] + [(f'chain growth p = {p}', p) for p in (0.98, 0.99, 0.995, 0.997, 0.999)]
for name, p in cases:
    net = synth.synth_network(n) if p is None else synth.synth_network_chain(n, p_chain=p, n_outlets=8, p_third=0.02)
    has = net.down_index >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
    indices = net.down_index[has].astype(np.int32)
    r = dt / net.k; den = r + 2 * (1 - net.x)
    c1, c2, c3 = (r - 2 * net.x) / den, (r + 2 * net.x) / den, (2 * (1 - net.x) - r) / den
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(-c1[indices], c2, c3, (c1 + c2) / dt)
        sch = plan.reserve(MODE_RAPID, T, 1)
        ti = plan.tile_info()
        stream = torch.cuda.current_stream().cuda_stream
        times = []
        for rep in range(3):
            q_t.zero_(); torch.cuda.synchronize(); t0 = time.perf_counter()
            plan.rapid_route_dev(q_t, ql, 288, out, 128, T, 1, stream)
            torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
        best = min(times[1:])
        print(f"{name:>24}: depth {plan.depth:>6}, headwaters {plan.n_headwaters:>7}, tiles {ti['tiles']} / {ti['levels']} levels, "
              f"{'k_tile' if sch['tiled'] else 'k_tick'} K = {sch['ticks_per_launch']}, ring {sch['ring_bytes'] / 1e9:.1f} GB: "
              f"{best * 1e3:.1f} ms, {n * T / best:.3e} reach-steps/s", flush=True)
    torch.cuda.empty_cache()

"""Bench line per network size on one GPU (random params order, permutation passes included) and the time of a one-month
hourly file (744 rows) at 1M reaches.    python profiles/microbench/size_sweep.py"""
import json, subprocess, sys, time
import numpy as np
sys.path.insert(0, '.')
for n, T in ((100_000, 35040), (250_000, 35040), (500_000, 35040), (1_000_000, 35040), (2_000_000, 17520), (4_000_000, 8760)):
    out = subprocess.run([sys.executable, 'bench.py', '--reaches', str(n), '--runoff-steps', str(T), '--steps', '2', '--warmup', '1', '--no-cpu-baseline', '--no-secondary'],
                         capture_output=True, text=True, env={**__import__('os').environ, 'RR_VERBOSE': '1'})
    line = [l for l in out.stdout.splitlines() if l.startswith('{')]
    rr = [l for l in out.stderr.splitlines() if l.startswith('rr:')]
    if not line:
        print(n, 'FAILED', out.stderr[-400:], flush=True)
        continue
    d = json.loads(line[-1])
    print(f"{n:>9} reaches x {T} rows: {d['value']:.3e} reach-steps/s, {d['ms_per_step']:.1f} ms, k_tile frac {d['roofline']['frac'] if d['roofline'] else None}; {rr[-1] if rr else ''}", flush=True)

from river_route_amd import synth
from river_route_amd.engine import Plan, DeviceBuffer, synchronize
n, T, rows = 1_000_000, 744, 48
net = synth.synth_network(n)
has = net.down_index >= 0
indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32); indices = net.down_index[has].astype(np.int32)
r = 900.0 / net.k; den = r + 2 * (1 - net.x)
c1, c2, c3 = (r - 2 * net.x) / den, (r + 2 * net.x) / den, (2 * (1 - net.x) - r) / den
with Plan(indptr, indices) as plan:
    plan.set_coeffs(-c1[indices], c2, c3, (c1 + c2) / 900.0)
    ql = synth.synth_qlateral(n, 0, rows)
    d_q, d_ql, d_out = DeviceBuffer(n * 8).upload(np.zeros(n)), DeviceBuffer(ql.nbytes).upload(ql), DeviceBuffer(rows * n * 8)
    for rep in range(4):
        synchronize(); t0 = time.perf_counter()
        plan.rapid_route_dev(d_q, d_ql, rows, d_out, rows, T, 1)
        synchronize(); dt = time.perf_counter() - t0
        print(f'744-row call at 1M reaches: {dt * 1e3:.2f} ms ({n * T / dt:.3e} reach-steps/s)', flush=True)

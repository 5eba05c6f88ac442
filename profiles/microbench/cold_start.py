"""Cold start of one plan (VERDICT r02 "what's weak" 7): rr_plan_create, rr_plan_set_coeffs, rr_plan_reserve, the first and a
steady one-month call (744 hourly rows, device arrays) at 100k and 1M reaches.    python profiles/microbench/cold_start.py"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from river_route_amd import synth
from river_route_amd.engine import Plan, DeviceBuffer, synchronize, MODE_RAPID

def clock(f):
    synchronize(); t0 = time.perf_counter(); out = f(); synchronize(); return (time.perf_counter() - t0) * 1e3, out

warm = DeviceBuffer(1 << 20); warm.free()      # the HIP context exists before anything is timed
for n in (100_000, 1_000_000):
    T, rows = 744, 48
    net = synth.synth_network(n)
    has = net.down_index >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32); indices = net.down_index[has].astype(np.int32)
    r = 900.0 / net.k; den = r + 2 * (1 - net.x)
    c1, c2, c3 = (r - 2 * net.x) / den, (r + 2 * net.x) / den, (2 * (1 - net.x) - r) / den
    t_create, plan = clock(lambda: Plan(indptr, indices))
    t_coef, _ = clock(lambda: plan.set_coeffs(-c1[indices], c2, c3, (c1 + c2) / 900.0))
    t_res, sch = clock(lambda: plan.reserve(MODE_RAPID, T, 1))
    t_res_year, sch_y = clock(lambda: plan.reserve(MODE_RAPID, 35_040, 1))
    ql = synth.synth_qlateral(n, 0, rows)
    d_q, d_ql, d_out = DeviceBuffer(n * 8).upload(np.zeros(n)), DeviceBuffer(ql.nbytes).upload(ql), DeviceBuffer(rows * n * 8)
    calls = [clock(lambda: plan.rapid_route_dev(d_q, d_ql, rows, d_out, rows, T, 1))[0] for _ in range(4)]
    print(f'{n:>8} reaches: rr_plan_create {t_create:7.1f} ms, rr_plan_set_coeffs {t_coef:6.1f} ms, rr_plan_reserve (744 rows: ring {sch["ring_bytes"] / 1e9:.2f} GB) {t_res:7.1f} ms, '
          f'again for a year ({sch_y["ring_bytes"] / 1e9:.2f} GB) {t_res_year:7.1f} ms; 744-row call: first {calls[0]:.2f} ms, then {calls[1]:.2f} / {calls[2]:.2f} / {calls[3]:.2f} ms '
          f'({n * T / (min(calls[1:]) * 1e-3):.3e} reach-steps/s)', flush=True)
    plan.close()

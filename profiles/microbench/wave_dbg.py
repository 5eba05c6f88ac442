"""Breakdown of one k_wave_rec launch from the per-block timestamps a debug build writes (development aid)."""
import numpy as np, sys
a = np.loadtxt(sys.argv[1])
t0 = a[:,1].min()
a[:,1:6] -= t0
a[:,1:6] /= 100.0   # us (100 MHz)
print("blocks", len(a), "launch span us", a[:,5].max())
start, pre, s1, s8, end = a[:,1], a[:,2], a[:,3], a[:,4], a[:,5]
def row(name, v):
    print(f"{name:22s} mean {v.mean():8.2f} p10 {np.percentile(v,10):8.2f} p50 {np.percentile(v,50):8.2f} p90 {np.percentile(v,90):8.2f} max {v.max():8.2f}")
for name, v in [("start", start), ("setup(start->loop)", pre-start), ("tick0 (rec wait)", s1-pre), ("ticks1-7", s8-s1), ("ticks8-15+state", end-s8), ("total", end-start), ("end", end)]:
    row(name, v)
sec = start >= 5
print("first-round blocks (start<5us):", (~sec).sum(), " second:", sec.sum())
if sec.any():
    for name, v in [("R2 start", start[sec]), ("R2 total", (end-start)[sec]), ("R1 total", (end-start)[~sec]), ("R1 setup", (pre-start)[~sec]), ("R2 setup", (pre-start)[sec]), ("R1 tick0", (s1-pre)[~sec]), ("R2 tick0", (s1-pre)[sec]), ("R1 t1-7", (s8-s1)[~sec]), ("R2 t1-7", (s8-s1)[sec]), ("R1 t8-15", (end-s8)[~sec]), ("R2 t8-15", (end-s8)[sec])]:
        row(name, v)

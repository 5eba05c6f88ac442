"""Per-tile phase breakdown of one k_tile launch from the timestamps a -DRR_WAVE_TRACE build writes (development aid).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRR_WAVE_TRACE river_route_amd/csrc/rr_plan.cpp \
        river_route_amd/csrc/rr_engine.hip -o /tmp/librr_trace.so
    RR_LIB_PATH=/tmp/librr_trace.so RR_WAVE_TRACE_DIAG=300 RR_WAVE_TRACE_FILE=/tmp/t.txt python bench.py --steps 1 --warmup 0 --no-cpu-baseline
    python profiles/microbench/wave_dbg.py /tmp/t.txt
"""
import sys
import numpy as np
a = np.loadtxt(sys.argv[1])
t = a[:, 1:15]
t0 = t[:, 0].min()
t = (t - t0) / 100.0   # us (100 MHz)
names = ['start', 'prologue (state)', 'first record', 'c0 issue + ticks 0-7', 'c0 store A', '-', 'c0 ticks 8-15', 'c0 store B',
         'c0 receive next', 'c1 issue + ticks 0-7', '-', 'c1 store A + ticks 8-15', 'rest of the loop', 'state out']
print('tiles', len(a), 'launch span us', t[:, 13].max())
first = t[:, 0] < 5
print('first round', first.sum(), 'second', (~first).sum())
for grp, m in (('R1', first), ('R2', ~first)):
    if not m.any():
        continue
    print(grp)
    prev = t[m, 0]
    print(f"  {'start':22s} mean {prev.mean():8.2f} max {prev.max():8.2f}")
    for k in range(1, 14):
        cur = t[m, k]
        ok = cur > 0
        if not ok.any():
            continue
        d = (cur - prev)[ok]
        print(f"  {names[k]:22s} mean {d.mean():8.2f} p10 {np.percentile(d, 10):8.2f} p90 {np.percentile(d, 90):8.2f} max {d.max():8.2f}   (at {cur[ok].mean():8.2f})")
        prev = np.where(ok, cur, prev)

if a.shape[1] >= 16:      # persistent workgroups: when each one left, relative to the first start of the launch
    end = (a[:, 15] - t0) / 100.0
    end = end[a[:, 15] > 0]
    print(f'workgroups {len(end)}: leave at mean {end.mean():.1f} us, p10 {np.percentile(end, 10):.1f}, p50 {np.percentile(end, 50):.1f}, '
          f'p90 {np.percentile(end, 90):.1f}, max {end.max():.1f}  -> the launch waits {end.max() - end.mean():.1f} us for its slowest workgroup')

"""Per-kernel medians of every counter in a rocprofv3 --pmc counter_collection.csv (one pass, any counters).
    python profiles/pmc_kernel_counters.py <counter_collection.csv> [kernel name substring ...]"""
import csv, re, statistics, sys
from collections import defaultdict
per = defaultdict(lambda: defaultdict(list))
with open(sys.argv[1], newline='') as f:
    for row in csv.DictReader(f):
        name = row['Kernel_Name']
        if len(sys.argv) > 2 and not any(k in name for k in sys.argv[2:]):
            continue
        m = re.search(r'(k_\w+(<[^>]*>)?)', name)
        short = m.group(1) if m else name[:60]
        per[short][row['Counter_Name']].append(float(row['Counter_Value']))
for k, counters in sorted(per.items()):
    n = len(next(iter(counters.values())))
    print(f'{k}  ({n} dispatches, medians)')
    for c, vals in sorted(counters.items()):
        print(f'    {c:<28} {statistics.median(vals):>16.1f}')

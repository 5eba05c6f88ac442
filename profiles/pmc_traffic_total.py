"""Reduce two rocprofv3 counter passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; separate runs of the same one-pass bench command) to the
HBM bytes the WHOLE timed pass moves, per kernel and per reach-step, and merge them into profiles/r05_pmc_traffic.json under the
params order they were taken on.

    python profiles/pmc_traffic_total.py <fetch counter_collection.csv> <write counter_collection.csv> --order random|postorder \
        --reaches 1000000 --runoff-steps 35040 [--main k_tile|k_direct] [--json profiles/r05_pmc_traffic.json]

gfx950 correction (/opt/skills/guides/MI355X_MICROARCH.md, HBM/rocprofv3 section): FETCH_SIZE counts half of a coalesced read
stream -> doubled; WRITE_SIZE exact.  Raw counters are KiB.  FETCH_SIZE is the L2's fabric-side request count: reads served by the
Infinity Cache are included, so it bounds HBM reads from above.  Every dispatch of the pass is summed (the command runs ONE pass:
--steps 1 --warmup 0 --no-cpu-baseline --no-secondary, so there is no parity-gate call in front of it); for the dominant kernel
the median over its full-grid dispatches is kept too (`main_kernel_bytes_per_full_launch`: what bench.py's roofline.traffic is).
The entry records the hash of the kernel sources; bench.py uses it only while that hash matches."""
import argparse, csv, hashlib, json, os, re, statistics, sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def engine_sha16():
    h = hashlib.sha256()
    csrc = os.path.join(REPO, 'river_route_amd', 'csrc')
    for name in sorted(f for f in os.listdir(csrc) if f.endswith(('.hip', '.hpp', '.cpp'))):
        with open(os.path.join(csrc, name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load(path, counter):
    per = defaultdict(list)
    with open(path, newline='') as f:
        for row in csv.DictReader(f):
            if row['Counter_Name'] != counter:
                continue
            m = re.search(r'(k_[a-z_0-9]+)', row['Kernel_Name'])
            per[m.group(1) if m and m.group(1) != 'k_copy16' else 'outside'].append((int(row['Grid_Size']), float(row['Counter_Value'])))      # torch's kernels (the forcing arrays) and the copy probe are not the timed pass
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('fetch_csv'); ap.add_argument('write_csv')
    ap.add_argument('--order', required=True, help="the entry's name: random | postorder (the headline's network), config2 | config4 | f32 (secondary lines)")
    ap.add_argument('--reaches', type=int, default=1_000_000)
    ap.add_argument('--runoff-steps', type=int, default=35_040)
    ap.add_argument('--main', default='k_tile')
    ap.add_argument('--command', default='')
    ap.add_argument('--json', default=os.path.join(REPO, 'profiles', 'r05_pmc_traffic.json'))
    a = ap.parse_args()
    fetch, write = load(a.fetch_csv, 'FETCH_SIZE'), load(a.write_csv, 'WRITE_SIZE')
    reach_steps = float(a.reaches) * a.runoff_steps
    kernels, total = {}, 0.0
    for k in sorted((set(fetch) | set(write)) - {'outside'}):
        rd = 2.0 * 1024.0 * sum(v for _, v in fetch.get(k, []))
        wr = 1024.0 * sum(v for _, v in write.get(k, []))
        if rd + wr < 1e-4 * reach_steps:      # state kernels, memsets: below a ten-thousandth of a byte per reach-step
            continue
        kernels[k] = {'dispatches': len(fetch.get(k, [])), 'hbm_read_bytes': rd, 'hbm_write_bytes': wr, 'bytes_per_reach_step': round((rd + wr) / reach_steps, 4)}
        total += rd + wr
    entry = {'engine_sha16': engine_sha16(), 'reaches': a.reaches, 'runoff_steps': a.runoff_steps, 'command': a.command,
             'note': 'FETCH_SIZE doubled (gfx950 counts half of a coalesced read stream), WRITE_SIZE as read; every dispatch of one pass summed',
             'bytes_per_reach_step': round(total / reach_steps, 3),
             'bytes_per_reach_step_by_kernel': {k: v['bytes_per_reach_step'] for k, v in kernels.items()}, 'kernels': kernels}
    if a.main in fetch:
        grid = max(g for g, _ in fetch[a.main])
        f = statistics.median(v for g, v in fetch[a.main] if g == grid)
        w = statistics.median(v for g, v in write[a.main] if g == grid)
        entry['main_kernel'] = a.main
        entry['main_kernel_bytes_per_full_launch'] = 2.0 * 1024.0 * f + 1024.0 * w
    try:
        with open(a.json) as fh:
            doc = json.load(fh)
    except (OSError, ValueError):
        doc = {}
    doc[a.order] = entry
    with open(a.json, 'w') as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps({a.order: {k: entry[k] for k in ('bytes_per_reach_step', 'bytes_per_reach_step_by_kernel')}}))


if __name__ == '__main__':
    main()

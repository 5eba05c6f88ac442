"""Reduce two rocprofv3 counter passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; separate runs of the same bench command) to HBM
bytes per dispatch of the routing kernel and the two record passes -> profiles/r03_pmc_traffic.json.

    python profiles/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> --positions 1035935 --ticks 64 \
        --reaches 1000000 > profiles/r03_pmc_traffic.json

gfx950 correction (/opt/skills/guides/MI355X_MICROARCH.md, HBM/rocprofv3 section): FETCH_SIZE counts half of a coalesced
read stream -> doubled; WRITE_SIZE exact.  Raw counters are KiB.  FETCH_SIZE is the L2's fabric-side request count: reads
served by the Infinity Cache are included, so it bounds HBM reads from above.  Only full-grid dispatches are used (the
pipeline launches partial grids while it fills and drains), medians over those.  The file records the hash of the kernel
sources it was taken with; bench.py uses it only while that hash matches.
"""
import argparse, csv, hashlib, json, os, statistics, sys
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def engine_sha16():
    h = hashlib.sha256()
    csrc = os.path.join(REPO, 'river_route_amd', 'csrc')
    for name in sorted(f for f in os.listdir(csrc) if f.endswith(('.hip', '.hpp', '.cpp'))):      # every source of librr_hip.so
        with open(os.path.join(csrc, name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load(path, counter):
    per = defaultdict(list)
    with open(path, newline='') as f:
        for row in csv.DictReader(f):
            if row['Counter_Name'] != counter:
                continue
            name = row['Kernel_Name']
            for short in ('k_tile<', 'k_rec_in', 'k_rec_out'):
                if short in name:
                    per[short.rstrip('<')].append((int(row['Grid_Size']), float(row['Counter_Value'])))
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('fetch_csv'); ap.add_argument('write_csv')
    ap.add_argument('--reaches', type=int, default=1_000_000)
    ap.add_argument('--positions', type=int, required=True, help='reaches + ghost positions (bench line: reaches + ghost_positions)')
    ap.add_argument('--rows-per-batch', type=int, default=128)
    ap.add_argument('--ticks', type=int, default=64)
    a = ap.parse_args()
    fetch, write = load(a.fetch_csv, 'FETCH_SIZE'), load(a.write_csv, 'WRITE_SIZE')
    out = {}
    for k in ('k_rec_in', 'k_rec_out', 'k_tile'):
        grid = max(g for g, _ in fetch[k])
        f = [v for g, v in fetch[k] if g == grid]
        w = [v for g, v in write[k] if g == grid]
        rd, wr = 2.0 * statistics.median(f) * 1024.0, statistics.median(w) * 1024.0
        e = {'grid_threads': grid, 'dispatches_sampled': len(f), 'FETCH_SIZE_KiB': statistics.median(f),
             'WRITE_SIZE_KiB': statistics.median(w), 'hbm_read_bytes': rd, 'hbm_write_bytes': wr, 'hbm_bytes': rd + wr}
        if k == 'k_tile':
            e.update(ticks_per_dispatch=a.ticks, positions=a.positions,
                     hbm_bytes_per_position_tick=round((rd + wr) / (a.positions * a.ticks), 6),
                     hbm_read_bytes_per_position_tick=round(rd / (a.positions * a.ticks), 6),
                     hbm_write_bytes_per_position_tick=round(wr / (a.positions * a.ticks), 6))
        else:
            e.update(rows_per_dispatch=a.rows_per_batch,
                     hbm_bytes_per_reach_row=round((rd + wr) / (a.reaches * a.rows_per_batch), 6))
        out[k] = e
    json.dump({'engine_sha16': engine_sha16(), 'command': 'rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --runoff-steps 6000',
               'note': 'FETCH_SIZE doubled (gfx950 counts half of a coalesced read stream), WRITE_SIZE as read; medians over full-grid dispatches',
               'kernels': out}, sys.stdout, indent=1)


if __name__ == '__main__':
    main()

"""Reduce two rocprofv3 counter passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; separate runs of the same bench command) to
HBM bytes per dispatch of the record-mode kernels -> profiles/r01_pmc_traffic.json['kernels'].

    python profiles/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv \
        --reaches 1000000 --rows-per-batch 128 > kernels.json

gfx950 correction (/opt/skills/guides/MI355X_MICROARCH.md, HBM/rocprofv3 section): FETCH_SIZE counts half of a coalesced
read stream -> doubled; WRITE_SIZE exact.  Raw counters are KiB.  Only full-grid dispatches are used (the time-tiled
pipeline launches partial grids while it fills and drains), medians over those.
"""
import argparse, csv, json, statistics, sys
from collections import defaultdict


def load(path, counter):
    per = defaultdict(list)
    with open(path, newline='') as f:
        for row in csv.DictReader(f):
            if row['Counter_Name'] != counter:
                continue
            name = row['Kernel_Name']
            for short in ('k_wave_rec', 'k_rec_in', 'k_rec_out'):
                if short in name:
                    per[short].append((int(row['Grid_Size']), float(row['Counter_Value'])))
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('fetch_csv'); ap.add_argument('write_csv')
    ap.add_argument('--reaches', type=int, default=1_000_000)
    ap.add_argument('--rows-per-batch', type=int, default=128)
    ap.add_argument('--ticks', type=int, default=16)
    a = ap.parse_args()
    fetch, write = load(a.fetch_csv, 'FETCH_SIZE'), load(a.write_csv, 'WRITE_SIZE')
    out = {}
    for k in ('k_rec_in', 'k_rec_out', 'k_wave_rec'):
        grid = max(g for g, _ in fetch[k])
        f = [v for g, v in fetch[k] if g == grid]
        w = [v for g, v in write[k] if g == grid]
        rd, wr = 2.0 * statistics.median(f) * 1024.0, statistics.median(w) * 1024.0
        e = {'grid_threads': grid, 'dispatches_sampled': len(f), 'FETCH_SIZE_KiB': statistics.median(f),
             'WRITE_SIZE_KiB': statistics.median(w), 'hbm_read_bytes': rd, 'hbm_write_bytes': wr, 'hbm_bytes': rd + wr}
        if k == 'k_wave_rec':
            e.update(ticks_per_dispatch=a.ticks, reaches=a.reaches,
                     hbm_bytes_per_reach_tick=round((rd + wr) / (a.reaches * a.ticks), 6))
        else:
            e.update(rows_per_dispatch=a.rows_per_batch,
                     hbm_bytes_per_reach_row=round((rd + wr) / (a.reaches * a.rows_per_batch), 6))
        out[k] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()

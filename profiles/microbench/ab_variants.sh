#!/bin/bash
# A/B on one box: bench line (ms per year, k_tile launch us) for each "LIB[:ENV=VAL[,ENV=VAL]]" argument, ROUNDS times round-robin.
#   bash profiles/microbench/ab_variants.sh "hip" "hip:RR_WAVE_K=128" "nt15" ...        (hip = the product library; an ENV only counts while csrc reads it:
#   RR_WAVE, RR_WAVE_K, RR_TILE_BLOCK, RR_TILE_LEAN, RR_UH_PAIRS, RR_DIRECT -- the round-3 knobs RR_REC_STREAM, RR_TILE_SLOTS, RR_WAVE_THREADS, RR_RING_EXTRA are gone)
# BENCH_ARGS adds bench.py arguments (e.g. BENCH_ARGS="--reaches 100000").  GATE=1: with the bench's parity gate against the oracle
# (a variant build is not covered by the test suite), a second of CPU baseline.
ROUNDS=${ROUNDS:-2}
for i in $(seq $ROUNDS); do
for spec in "$@"; do
  lib=${spec%%:*}; envs=""; [[ "$spec" == *:* ]] && envs=${spec#*:}
  path=$PWD/gpurun_variants/librr_$lib.so; [ "$lib" = hip ] && path=$PWD/river_route_amd/librr_hip.so
  echo -n "$spec: "
  gate="--no-cpu-baseline"; [ -n "$GATE" ] && gate="--cpu-baseline-seconds 1 --cpu-replicas 0"
  env RR_LIB_PATH=$path ${envs//,/ } timeout -k 10 300 python bench.py $BENCH_ARGS --steps 3 --warmup 1 --no-secondary $gate 2>/dev/null | python -c "
import sys, json
ls = [l for l in sys.stdin if l.startswith('{')]
if not ls: print('FAILED')
else:
    d = json.loads(ls[-1]); r = d['roofline'] or {}
    print('%.2f ms  %.4g reach-steps/s  k_tile %.1f us frac %.3f  %s' % (d['ms_per_step'], d['value'], r.get('avg_launch_us', 0), r.get('frac', 0), (d.get('cpu_baseline') or {}).get('parity_gate', '')[-52:]))"
done; done

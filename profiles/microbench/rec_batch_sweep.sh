# record-pass batch size: the default build (RR_REC_BATCH 8) against a variant (hipcc ... -DRR_REC_BATCH=16 -DRR_REC_COLS=32
# -o river_route_amd/librr_b16c32.so); the bench's
# parity gate runs for each; default forcing (288 rows) and round 1's rings (96 / 96)
for i in 1 2; do
for lib in librr_hip.so librr_b16c32.so; do
  for args in "" "--forcing-rows 96 --sink-rows 96"; do
  echo -n "$lib $args: "
  RR_LIB_PATH=$PWD/river_route_amd/$lib timeout -k 10 300 python bench.py $args --steps 3 --warmup 1 --cpu-replicas 0 --cpu-baseline-seconds 2 2>&1 | python -c "
import sys,json
ls=[l for l in sys.stdin if l.startswith('{')]
if not ls: print('FAILED')
else:
    d=json.loads(ls[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['cpu_baseline']['parity_gate'][-40:])"
  done
done; done

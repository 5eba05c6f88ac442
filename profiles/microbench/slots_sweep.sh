# k_tile launch time at 1M reaches against the number of workgroups a launch is given (RR_TILE_SLOTS x 1024 / 512):
# 256 = one resident set of persistent workgroups, more = the dispatcher refills slots as workgroups leave.
for p in 256 320 384 512 768 1024; do
  echo "== RR_TILE_SLOTS=$p"
  RR_TILE_SLOTS=$p timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_launch_us'))"
done

# A/B of two builds of the engine on one box (boxes differ by +-3 %): river_route_amd/librr_prev.so is an earlier commit's
# sources built with the hipcc line of river_route_amd/_lib.py (git show <commit>:river_route_amd/csrc/... into a scratch directory).
for i in 1 2 3; do
for lib in librr_prev.so librr_hip.so; do
  for args in "" "--reaches 100000" "--workload unit"; do
  echo -n "$lib $args: "
  RR_LIB_PATH=$PWD/river_route_amd/$lib timeout -k 10 200 python bench.py $args --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
  done
done; done

#!/bin/bash
# Collects the round-2 profiles on the GPU box (run from the repo root through gpurun; everything lands in gpurun_out/r02/,
# the summaries are then copied to profiles/ by hand).  Counter passes run apart from the kernel trace.
#   bash profiles/collect_r02.sh stats     kernel summaries a / b / c
#   bash profiles/collect_r02.sh pmc       FETCH_SIZE and WRITE_SIZE passes + reduction
#   bash profiles/collect_r02.sh bench     default bench line and the config-4 line
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $REPO
case "$1" in
stats)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/a.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b -- python3 bench.py --workload unit --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c -- python3 bench.py --substeps 4 --runoff-steps 8760 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/c.log 2>&1
  for k in a b c; do cp $(find $OUT/$k -name '*kernel_stats.csv' | head -1) $OUT/${k}_kernel_stats.csv; rm -rf $OUT/$k; done
  ;;
pmc)
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --runoff-steps 6000 > $OUT/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --runoff-steps 6000 > $OUT/write.log 2>&1
  python3 profiles/pmc_traffic.py $(find $OUT/fetch -name '*counter_collection.csv' | head -1) $(find $OUT/write -name '*counter_collection.csv' | head -1) \
      --positions 1035935 --ticks 64 --reaches 1000000 > $OUT/pmc_traffic.json
  rm -rf $OUT/fetch $OUT/write
  ;;
bench)
  python3 bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err
  python3 bench.py --workload unit > $OUT/bench_unit_config4.log 2> $OUT/bench_unit_config4.err
  ;;
*) echo "usage: $0 stats|pmc|bench"; exit 2;;
esac

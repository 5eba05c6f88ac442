#!/bin/bash
# Collects the round-3 profiles on the GPU box (run from the repo root through gpurun; everything lands in gpurun_out/r03/,
# the summaries are then copied to profiles/ by hand).  Counter passes run apart from the kernel trace.
#   bash profiles/collect_r03.sh stats     kernel summaries a / b / c
#   bash profiles/collect_r03.sh pmc       FETCH_SIZE and WRITE_SIZE passes + reduction
#   bash profiles/collect_r03.sh bench     default bench line and the config-4 line
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $REPO
case "$1" in
stats)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/a.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b -- python3 bench.py --workload unit --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c -- python3 bench.py --substeps 4 --runoff-steps 8760 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/c.log 2>&1
  for k in a b c; do cp $(find $OUT/$k -name '*kernel_stats.csv' | head -1) $OUT/${k}_kernel_stats.csv; rm -rf $OUT/$k; done
  ;;
counters)
  # what binds the fused convolution pass (and the other kernels of BASELINE config 4): SQ counters, one pass
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -- python3 bench.py --workload unit --steps 1 --warmup 0 --no-cpu-baseline > $OUT/sq.log 2>&1
  python3 profiles/pmc_kernel_counters.py $(find $OUT/sq -name '*counter_collection.csv' | head -1) k_rec_in_uh k_tile k_rec_out > $OUT/b_sq_counters.txt
  rm -rf $OUT/sq
  ;;
pmc)
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --runoff-steps 6000 > $OUT/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --runoff-steps 6000 > $OUT/write.log 2>&1
  python3 profiles/pmc_traffic.py $(find $OUT/fetch -name '*counter_collection.csv' | head -1) $(find $OUT/write -name '*counter_collection.csv' | head -1) \
      --positions 1035935 --ticks 64 --reaches 1000000 > $OUT/pmc_traffic.json
  rm -rf $OUT/fetch $OUT/write
  ;;
bench)
  python3 bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err
  python3 bench.py --workload unit > $OUT/bench_unit_config4.log 2> $OUT/bench_unit_config4.err
  ;;
*) echo "usage: $0 stats|pmc|counters|bench"; exit 2;;
esac

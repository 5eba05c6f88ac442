#!/bin/bash
# Collects the round-5 profiles on the GPU box (run from the repo root through gpurun; everything lands in gpurun_out/r05/, the
# summaries are then copied to profiles/ by hand).  Counter passes run apart from the kernel trace.
#   bash profiles/collect_r05.sh stats     kernel summaries: a headline (random order), d the same year in post-order (direct row path), b config 4, f float32 rows, u config 4 in post-order
#   bash profiles/collect_r05.sh pmc [a|b] FETCH_SIZE and WRITE_SIZE passes over the whole timed pass of: (a) the headline in both orders, config 2, (b) config 4 in both orders, the float32 line
#                                          -> gpurun_out/r05/r05_pmc_traffic.json (no secondary line of bench.py without its own traffic)
#   bash profiles/collect_r05.sh bench     the default bench line
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $REPO
ONE="--steps 1 --warmup 0 --no-cpu-baseline --no-secondary"
case "$1" in
stats)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/a.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d -- python3 bench.py --order postorder --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/d.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b -- python3 bench.py --workload unit --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/f -- python3 bench.py --workload rapid_f32 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/f.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/u -- python3 bench.py --workload unit --order postorder --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/u.log 2>&1
  for k in a d b f u; do cp $(find $OUT/$k -name '*kernel_stats.csv' | head -1) $OUT/${k}_kernel_stats.csv; rm -rf $OUT/$k; done
  ;;
pmc)
  cp profiles/r05_pmc_traffic.json $OUT/r05_pmc_traffic.json 2>/dev/null || true
  pass() {      # name, main kernel, reaches, rows, bench arguments
    local name=$1 main=$2 reaches=$3 rows=$4; shift 4
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py "$@" $ONE > $OUT/fetch_$name.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py "$@" $ONE > $OUT/write_$name.log 2>&1
    python3 profiles/pmc_traffic_total.py $(find $OUT/fetch -name '*counter_collection.csv' | head -1) $(find $OUT/write -name '*counter_collection.csv' | head -1) \
        --order $name --main $main --reaches $reaches --runoff-steps $rows --json $OUT/r05_pmc_traffic.json --command "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py $* $ONE"
    rm -rf $OUT/fetch $OUT/write
  }
  case "${2:-all}" in      # (two gpurun calls of at most 20 minutes: `pmc a`, then `pmc b` with gpurun_out/r05/r05_pmc_traffic.json copied to profiles/ in between)
  a|all)
    pass random k_tile 1000000 35040 --order random
    pass postorder k_direct 1000000 35040 --order postorder
    pass config2 k_tile 100000 35040 --reaches 100000
    ;;&
  b|all)
    pass config4 k_tile 1000000 3504 --workload unit
    pass config4_postorder k_direct 1000000 3504 --workload unit --order postorder
    pass f32 k_tile 1000000 35040 --workload rapid_f32
    ;;
  esac
  ;;
bench)
  python3 bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err
  ;;
*) echo "usage: $0 stats|pmc|bench"; exit 2;;
esac

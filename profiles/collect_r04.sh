#!/bin/bash
# Collects the round-4 profiles on the GPU box (run from the repo root through gpurun; everything lands in gpurun_out/r04/, the
# summaries are then copied to profiles/ by hand).  Counter passes run apart from the kernel trace.
#   bash profiles/collect_r04.sh stats     kernel summaries: a headline (random order), d the same year in post-order (direct row path), b config 4
#   bash profiles/collect_r04.sh pmc       FETCH_SIZE and WRITE_SIZE passes over the whole year, both orders -> gpurun_out/r04/r04_pmc_traffic.json
#   bash profiles/collect_r04.sh bench     the default bench line
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $REPO
ONE="--steps 1 --warmup 0 --no-cpu-baseline --no-secondary"
case "$1" in
stats)
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/a -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/a.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/d -- python3 bench.py --order postorder --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/d.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b -- python3 bench.py --workload unit --steps 2 --warmup 1 --no-cpu-baseline > $OUT/b.log 2>&1
  for k in a d b; do cp $(find $OUT/$k -name '*kernel_stats.csv' | head -1) $OUT/${k}_kernel_stats.csv; rm -rf $OUT/$k; done
  ;;
pmc)
  cp profiles/r04_pmc_traffic.json $OUT/r04_pmc_traffic.json 2>/dev/null || true
  for order in random postorder; do
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --order $order $ONE > $OUT/fetch_$order.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --order $order $ONE > $OUT/write_$order.log 2>&1
    main=k_tile; [ $order = postorder ] && main=k_direct
    python3 profiles/pmc_traffic_total.py $(find $OUT/fetch -name '*counter_collection.csv' | head -1) $(find $OUT/write -name '*counter_collection.csv' | head -1) \
        --order $order --main $main --json $OUT/r04_pmc_traffic.json --command "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --order $order $ONE"
    rm -rf $OUT/fetch $OUT/write
  done
  ;;
bench)
  python3 bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err
  ;;
*) echo "usage: $0 stats|pmc|bench"; exit 2;;
esac

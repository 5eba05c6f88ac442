#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of the headline's kernels for each "LIB[:ENV=VAL,...]" argument (see kernel_times.sh), 6,000 rows:
#   bash profiles/microbench/pmc_write.sh OUTDIR hip whole
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
k=0
for spec in "$@"; do
  k=$((k+1))
  lib=${spec%%:*}; envs=""; [[ "$spec" == *:* ]] && envs=${spec#*:}
  path=$PWD/gpurun_variants/librr_$lib.so; [ "$lib" = hip ] && path=$PWD/river_route_amd/librr_hip.so
  for c in FETCH_SIZE WRITE_SIZE; do
    ( export RR_LIB_PATH=$path; for kv in ${envs//,/ }; do export $kv; done
      rocprofv3 --pmc $c --output-format csv -d $out/p$k$c -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --runoff-steps 6000 > $out/p$k$c.log 2>&1 )
  done
  echo "== $spec"
  python3 profiles/pmc_traffic.py $(find $out/p${k}FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find $out/p${k}WRITE_SIZE -name '*counter_collection.csv' | head -1) \
      --positions 1035935 --ticks 64 --reaches 1000000 | python3 -c "
import json, sys
d = json.load(sys.stdin)['kernels']
for k, v in d.items():
    print('   %-10s read %8.1f MB  write %8.1f MB per dispatch' % (k, v['hbm_read_bytes'] / 1e6, v['hbm_write_bytes'] / 1e6))
"
  rm -rf $out/p${k}FETCH_SIZE $out/p${k}WRITE_SIZE
done

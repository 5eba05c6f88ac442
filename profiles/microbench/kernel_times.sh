#!/bin/bash
# Average duration of the three kernels of the headline under rocprofv3 for each "LIB[:ENV=VAL,...][@bench args]" argument:
#   bash profiles/microbench/kernel_times.sh OUTDIR "hip" "alias:RR_ALIAS=1" "hip@--forcing-rows 1 --sink-rows 1"
# rocprofv3 gets the program itself after `--` (no env/bash hop): the variables are exported in this shell first.
out=$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
k=0
for spec in "$@"; do
  k=$((k+1))
  args=""; [[ "$spec" == *@* ]] && args=${spec#*@}; spec=${spec%%@*}
  lib=${spec%%:*}; envs=""; [[ "$spec" == *:* ]] && envs=${spec#*:}
  path=$PWD/gpurun_variants/librr_$lib.so; [ "$lib" = hip ] && path=$PWD/river_route_amd/librr_hip.so
  ( export RR_LIB_PATH=$path; for kv in ${envs//,/ }; do export $kv; done
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/p$k -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary $args > $out/p$k.log 2>&1 )
  f=$(find $out/p$k -name '*kernel_stats.csv' | head -1)
  echo "== $spec $args: $(grep -o '"ms_per_step": [0-9.]*' $out/p$k.log | tail -1)"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if any(k in n for k in ('k_tile<', 'k_rec_in', 'k_rec_out', 'k_copy')):
        print('   %-60s calls %6s avg %9.1f us  %5s %%' % (n[:60], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
  rm -rf $out/p$k
done

# per-kernel summary of leaf_part_probe.py cases under two allocation shifts
set -e
REPO=$PWD; OUT=$REPO/gpurun_out/leaf; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $REPO
for sh in 0 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b$sh -- python3 profiles/microbench/leaf_part_probe.py b $sh > $OUT/b$sh.log 2>&1
  cp $(find $OUT/b$sh -name '*kernel_stats.csv' | head -1) $OUT/b${sh}_kernel_stats.csv; rm -rf $OUT/b$sh
  echo == shift $sh; grep "^(" $OUT/b$sh.log; head -4 $OUT/b${sh}_kernel_stats.csv | cut -d, -f1-4
done

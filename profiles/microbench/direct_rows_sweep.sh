# k_direct with the caller's rows in the Infinity Cache (4-row rings) and in HBM (1,024-row rings): is a tick bound by memory at all?
mkdir -p gpurun_out/r04
for rs in "4 4" "1024 1024" "4 1024" "1024 4"; do
  echo "rows sink = $rs"
  timeout -k 10 120 python profiles/microbench/direct_time.py 1000000 35040 $rs 1 2>&1 | grep postorder | cut -c1-160 || exit 1
done

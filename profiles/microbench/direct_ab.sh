# A/B of librr_hip.so builds on one box: direct_ab.sh name [name ...]  (river_route_amd/_ab/<name>.so, built by hand from variants of csrc)
# ROWS="forcing sink" sets the rings (default 1024 1024)
mkdir -p gpurun_out/r04
cp river_route_amd/librr_hip.so /tmp/librr_hip.keep.so
for v in "$@"; do
  cp river_route_amd/_ab/$v.so river_route_amd/librr_hip.so
  echo "== $v"
  timeout -k 10 120 python profiles/microbench/direct_time.py 1000000 35040 ${ROWS:-1024 1024} 2 2>&1 | grep postorder | cut -c1-150 || exit 1
done
cp /tmp/librr_hip.keep.so river_route_amd/librr_hip.so

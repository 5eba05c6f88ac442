# A/B of librr_hip.so builds on one box: direct_ab.sh name [name ...]  (river_route_amd/_ab/<name>.so, built by hand from variants of csrc;
# RR_LIB_PATH selects the library, nothing is copied).  ROWS="forcing sink" sets the rings (default 1024 1024), T the rows (default 35040).
for v in "$@"; do
  echo "== $v"
  RR_LIB_PATH=$PWD/river_route_amd/_ab/$v.so timeout -k 10 150 python profiles/microbench/direct_time.py 1000000 ${T:-35040} ${ROWS:-1024 1024} 2 2>&1 | grep postorder | tail -1 | cut -c1-260 || exit 1
done

# out-pass / in-pass column tiles in XCD-contiguous order (RR_REC_SWIZZLE bit 1 / bit 0): a 1.25M-reach part with an odd column
# count (b), a connected 1.25M-reach network (c), and the 1M bench
for k in 0 2 0 2; do echo "RR_REC_SWIZZLE=$k bench 1M"; RR_REC_SWIZZLE=$k timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])"; done
for k in 0 1 2 3; do echo RR_REC_SWIZZLE=$k; RR_REC_SWIZZLE=$k timeout -k 10 200 python profiles/microbench/leaf_part_probe.py bc 0 128 2>&1 | grep "^("; done

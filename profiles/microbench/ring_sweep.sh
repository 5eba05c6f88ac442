for x in 0 120 0 120 7 33; do
  echo "== RR_RING_EXTRA=$x"
  RR_RING_EXTRA=$x timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_launch_us'), d['roofline']['peak_measured_copy'])"
done

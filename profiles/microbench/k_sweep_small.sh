for n in 100000 250000 500000; do for k in 64 128 256; do
  echo -n "n=$n K=$k: "
  RR_WAVE_K=$k timeout -k 10 200 python bench.py --reaches $n --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"
done; done

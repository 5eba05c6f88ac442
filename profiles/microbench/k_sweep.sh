for k in 32 64 128 256; do
  echo "== RR_WAVE_K=$k"
  RR_WAVE_K=$k RR_VERBOSE=1 timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/k$k.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_launch_us'), d['roofline']['ticks_per_launch'])"
  grep "^rr:" gpurun_out/k$k.err | tail -1
done

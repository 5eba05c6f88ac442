# A/B (round 5): rows per direct task (RR_WAVE_K) against ticks per skeleton task, post-order network.  RR_DIRECT_KS was a temporary knob of the build
# this ran on (choose_schedule read it); the outcome is the rule direct_KS() in rr_exec.hpp and profiles/r05_direct_ks_ab.txt -- kept as the record of the command.
#   bash profiles/microbench/direct_ks_ab.sh [bench arguments, e.g. --runoff-steps 3504]   with SPECS="K:KS ..." (KS empty: the default)
for spec in ${SPECS:-512: 1024:512 1024:256 2048:512 512:256 512:}; do
  K=${spec%%:*}; KS=${spec#*:}
  if [ -n "$KS" ]; then export RR_DIRECT_KS=$KS; else unset RR_DIRECT_KS; fi
  RR_WAVE_K=$K python bench.py --order postorder --steps 2 --warmup 1 --no-cpu-baseline --no-secondary "$@" > gpurun_out/ks_run.log 2>&1
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/ks_run.log") if l.startswith("{")][-1])
k=d["roofline"]["path"]["kernels"]
print("K=$K KS=$KS", round(d["ms_per_step"],2), {n:(v["launches"], v["avg_us"]) for n,v in k.items()})
PY
done

"""Bench line per network size and tile size (RR_WAVE_THREADS = positions per tile = threads per workgroup of k_tile): which tile
size the engine should pick by itself (rr_plan_create).    python profiles/microbench/threads_sweep.py [n ...]"""
import json, os, subprocess, sys
sizes = [int(a) for a in sys.argv[1:]] or [100_000, 250_000, 500_000]
for n in sizes:
    for th in ('auto', '128', '256', '512'):
        env = {**os.environ, 'RR_VERBOSE': '1'}
        env.pop('RR_WAVE_THREADS', None)
        if th != 'auto':
            env['RR_WAVE_THREADS'] = th
        out = subprocess.run([sys.executable, 'bench.py', '--reaches', str(n), '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-secondary'],
                             capture_output=True, text=True, env=env)
        line = [l for l in out.stdout.splitlines() if l.startswith('{')]
        rr = [l for l in out.stderr.splitlines() if l.startswith('rr:')]
        if not line:
            print(n, th, 'FAILED', out.stderr[-300:], flush=True)
            continue
        d = json.loads(line[-1])
        r = d['roofline'] or {}
        print(f"{n:>8} reaches, tile {th:>4}: {d['value']:.3e} reach-steps/s, {d['ms_per_step']:.1f} ms, k_tile {r.get('avg_launch_us')} us frac {r.get('frac')}; {rr[-1] if rr else ''}", flush=True)

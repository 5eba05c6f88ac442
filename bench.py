"""
bench.py -- reach-steps/sec of the Muskingum routing hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W

Workload at N=1 (BASELINE.json configs[2]): RapidMuskingum on the 1M-reach synthetic network, 1 year at
15-minute steps (35,040 runoff steps, dt_routing = dt_runoff = 900 s, fp64).  One bench "step" is ONE pass of
the hot path over that year: a single rr_rapid_route_dev call.  A year of lateral inflow for 1M reaches is
280 GB, so the forcing is a 288-row (three days) device-resident array read cyclically (row t % 288) and the
discharge goes to a 128-row cyclic sink.  Both are at least as long as what one launch of the record passes moves (143
rows in, 128 out), so that -- as with a real T-row array -- no launch reads or writes a row twice and finds it in a
cache (SURVEY section 8d proposes 96-row rings; with those a quarter of the rows of every launch never reached HBM);
every routed row is read from and written to HBM, and the
params-order <-> engine-order permutation passes are inside the timed region.  Inputs are resident in HBM
when the timed region starts (the PCIe-inclusive host-pointer rate is noted in DESIGN.md, never here).

The JSON line carries `roofline` for the dominant kernel (the time-tiled routing kernel, k_tile) from HIP events
recorded by the engine around sampled launches on its own stream -- algorithmic bytes of the time-tiled step over the
launch time, a fraction of the 8 TB/s peak by construction, with the measured device-copy rate beside it (DESIGN.md
section 5) -- and `cpu_baseline`: the oracle (oracle/rr_oracle.c, -O3 -march=native -ffast-math, 1 thread -- the
reference path is single-threaded by construction) timed on a bounded sample of the same workload on this box's host
cores.  Before anything is timed the result of the SAME kernels (time-tiled kernel + record permutation passes, forced)
is compared with the oracle's on every row of the forcing; a mismatch refuses to report.

N > 1 (one rank per GPU, torch.distributed.run): BASELINE config 5 -- ONE network of 1.25M reaches per GPU (10M at
N = 8) graph-partitioned over the GPUs, boundary discharge exchanged over RCCL (river_route_amd/multi_gpu.py).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # before the HIP runtime starts: RCCL's device-memory sharing needs dmabuf IPC on these hosts

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--reaches', type=int, default=None, help='reaches per GPU (default 1,000,000 at N = 1; 1,250,000 at N > 1: BASELINE config 5 is 10M reaches on 8 GPUs)')
    ap.add_argument('--runoff-steps', type=int, default=35_040, help='runoff steps per bench step (1 yr @ 15 min)')
    ap.add_argument('--substeps', type=int, default=1)
    ap.add_argument('--forcing-rows', type=int, default=0,
                    help='rows of the cyclic device-resident forcing; 0 = 288 (three days: more than the 143 rows one in-pass launch reads, so no launch reads a '
                         'forcing row twice), 1,024 on the direct row path (8 GB: a tile comes back to a row after the chip has streamed the whole array)')
    ap.add_argument('--sink-rows', type=int, default=0,
                    help='rows of the cyclic discharge sink; 0 = one out-pass launch (128), so that no launch writes a row twice; 1,024 on the direct row path')
    ap.add_argument('--order', default='random', choices=['random', 'levels', 'bfs', 'postorder'],
                    help="params-file order of the synthetic network; 'postorder' (depth-first post-order: every sub-basin a run of consecutive "
                         "columns) lets the engine route straight from and to the rows (direct row path, DESIGN.md section 3d)")
    ap.add_argument('--sample-every', type=int, default=128)
    ap.add_argument('--chunk-rows', type=int, default=16)
    ap.add_argument('--cpu-baseline-steps', type=int, default=96)
    ap.add_argument('--cpu-baseline-seconds', type=float, default=12.0)
    ap.add_argument('--exchange-rows', type=int, default=128,
                    help='N>1: runoff steps per boundary-series message')
    ap.add_argument('--workload', default='rapid', choices=['rapid', 'unit', 'rapid_f32', 'dropin'],
                    help="'unit' = BASELINE config 4 (UnitMuskingum + 48-step UH kernel), 'rapid_f32' = the headline's year with float32 rows in and "
                         "hourly float32 means out; secondary lines, not the headline (run alone for the counter passes)")
    ap.add_argument('--uh-steps', type=int, default=48)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--sequential-parts', type=int, default=0,
                    help='one GPU: the network of `--gpus P` (P x --reaches reaches, BASELINE config 5 at P = 8) cut into P parts that are routed '
                         'one after another on this card, boundary series handed from part to part (multi_gpu.run_sequential); per-part times in the line')
    ap.add_argument('--no-secondary', action='store_true',
                    help="N = 1: skip the `secondary` entries (BASELINE configs 2 and 4 behind their own oracle gates)")
    ap.add_argument('--cpu-replicas', type=int, default=-1,
                    help="also time N independent oracle replicas on N cores (the only parallelism the reference endorses, "
                         "docs/references/parallelism.md:67-114); 0 = off, -1 = a sweep over 16 / 64 / 256 / all cores")
    a = ap.parse_args()
    if a.reaches is None:
        a.reaches = 1_000_000 if a.gpus == 1 and int(os.environ.get('WORLD_SIZE', '1')) == 1 and a.sequential_parts < 2 else 1_250_000
    return a


def csc_from_down(down_index):
    has = down_index >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
    return indptr, down_index[has].astype(np.int32)


def muskingum_coefficients(k, x, dt):
    """river_route/routers/Muskingum.py:174-179."""
    r = dt / k
    den = r + 2.0 * (1.0 - x)
    return (r - 2.0 * x) / den, (r + 2.0 * x) / den, (2.0 * (1.0 - x) - r) / den


def host_cpu_model():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown CPU'


def cpu_baseline(net, indptr, indices, c1, c2, c3, dt, nsub, steps, seconds, replicas=0):
    """Oracle on the host: same network, first `steps` runoff steps of the same forcing."""
    from oracle import oracle
    from river_route_amd import synth
    import tempfile
    n = net.n
    out_dir = tempfile.mkdtemp(prefix='rr_oracle_')
    oracle.build(fast=True, out_dir=out_dir)          # compiled here for this box's CPU
    ql = synth.synth_qlateral(n, 0, steps, dt=dt * nsub)
    lhs = -c1[indices]
    c4_dt = (c1 + c2) / (dt * nsub)
    q, d = np.zeros(n), np.zeros((steps, n))
    oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q, ql[:2], d[:2], nsub, fast=True, out_dir=out_dir)
    q[:] = 0.0
    first_pass = None
    reps, dt_s = 0, 0.0
    while dt_s < seconds and reps < 64:      # the same `steps` rows again and again, state carried over
        t0 = time.perf_counter()
        oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q, ql, d, nsub, fast=True, out_dir=out_dir)
        dt_s += time.perf_counter() - t0
        reps += 1
        if first_pass is None:
            first_pass = d.copy()       # zero initial state, rows [0, steps): what the GPU self-check must reproduce
    out = {'value': n * steps * nsub * reps / dt_s, 'unit': 'reach-steps/s', 'cores': 1, 'kind': 'port',
           'sample': f'{n} reaches x {steps * reps} runoff steps ({reps} passes over {steps} forcing rows) x {nsub} '
                     f'sub-step(s), {dt_s:.2f} s, oracle/rr_oracle.c gcc -O3 -march=native -ffast-math, 1 thread '
                     f'of {os.cpu_count()} host cores ({host_cpu_model()})',
           'first_pass': first_pass}
    if replicas:
        out['replicas'] = cpu_replicas(replicas, indptr, indices, lhs, c2, c3, c4_dt, ql, nsub, out_dir, seconds)
    return out


def _replica_worker(args):
    """One independent oracle replica pinned to nothing in particular: the OS spreads the processes over the cores."""
    indptr, indices, lhs, c2, c3, c4_dt, ql, nsub, out_dir, seconds = args
    from oracle import oracle
    n, steps = c2.shape[0], ql.shape[0]
    q, d = np.zeros(n), np.zeros((steps, n))
    reps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q, ql, d, nsub, fast=True, out_dir=out_dir)
        reps += 1
    return n * steps * nsub * reps, time.perf_counter() - t0


def cpu_replicas(replicas, indptr, indices, lhs, c2, c3, c4_dt, ql, nsub, out_dir, seconds):
    """N independent replicas on N cores (docs/references/parallelism.md:67-114: one process per ensemble member or
    watershed is the only parallelism the reference endorses), at several widths up to the box's core count: the line shows
    where the host saturates (every replica streams its own 1M-reach arrays through the memory system) and reports the best
    aggregate.  Shorter forcing so N copies fit host memory; before this process touches the GPU (the replicas fork)."""
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    widths = sorted({min(w, cores) for w in ((16, 64, 256, cores) if replicas < 0 else (replicas,))})
    rows = ql[:min(ql.shape[0], 8)]
    sweep = []
    for N in widths:
        with mp.get_context('fork').Pool(N) as pool:
            res = pool.map(_replica_worker, [(indptr, indices, lhs, c2, c3, c4_dt, rows, nsub, out_dir, min(seconds, 3.0))] * N)
        sweep.append({'replicas': N, 'value': sum(r[0] for r in res) / max(r[1] for r in res)})
    best = max(sweep, key=lambda e: e['value'])
    return {'value': best['value'], 'unit': 'reach-steps/s', 'cores': best['replicas'], 'host_cores': cores, 'sweep': sweep,
            'sample': f'independent single-thread replicas of the same network, {rows.shape[0]} forcing rows each, ~3 s per width; best of {[e["replicas"] for e in sweep]}'}


def bench_unit(args, device_index):
    """BASELINE config 4: UnitMuskingum, 1M reaches, 48-step UH kernel.  One bench step = convolve + route one
    block of runoff depths (default 3,504 steps = 1/10 year; the (T, n) depth block must be resident for the
    convolution, 28 GB at 1M reaches)."""
    import torch
    from river_route_amd import synth
    from river_route_amd.engine import Plan
    n, nsub, dt, n_ks = args.reaches, args.substeps, 900.0, args.uh_steps
    T = args.runoff_steps if args.runoff_steps != 35_040 else 3_504
    dev = torch.device('cuda', device_index)
    net = synth.synth_network(n, order=args.order)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = muskingum_coefficients(net.k, net.x, dt / nsub)
    plan = Plan(indptr, indices, device=device_index)
    plan.set_coeffs(-c1[indices], c2, c3, None)
    plan.set_options(rows_per_chunk=args.chunk_rows, sample_every=args.sample_every)
    kern_h = synth.synth_uh_kernel(n, n_ks, tr=dt)
    kern = torch.from_numpy(kern_h).to(dev)
    state = torch.zeros_like(kern)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    depth = torch.rand((T, n), dtype=torch.float64, device=dev, generator=g) * 1e-3
    out = torch.zeros((T, n), dtype=torch.float64, device=dev)
    n_inner = plan.n_inner
    q_ch = torch.zeros(n_inner, dtype=torch.float64, device=dev)
    q_full = torch.zeros(n_inner, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def one_pass():      # UnitMuskingum._router of one file: convolution fused into the in-pass, routing, state bookkeeping
        state.zero_(); q_ch.zero_(); q_full.zero_()
        plan.unit_route_uh_dev(q_ch, q_full, None, kern, state, n_ks, depth, T, nsub, discharge=out, stream=stream)

    base = None
    if not args.no_cpu_baseline:
        from oracle import oracle
        import tempfile
        out_dir = tempfile.mkdtemp(prefix='rr_oracle_')
        oracle.build(fast=True, out_dir=out_dir)
        Tc = 64
        d_h = depth[:Tc].cpu().numpy()
        from tests_support import unit_split_arrays
        hw_idx, inner_idx, A_in, A_hw = unit_split_arrays(indptr, indices, n)
        c1i, c2i, c3i = c1[inner_idx], c2[inner_idx], c3[inner_idx]
        uh = oracle.UnitHydrograph(kern_h)
        t0 = time.perf_counter()
        conv_h = uh.convolve(d_h)
        qc, qf, dd = np.zeros(inner_idx.size), np.zeros(inner_idx.size), np.zeros((Tc, n))
        oracle.unit_route(A_in.indptr, A_in.indices, -c1i[A_in.indices], A_in.indptr, A_in.indices, A_in.data,
                          A_hw.indptr, A_hw.indices, A_hw.data, c1i, c2i, c3i, hw_idx, inner_idx, qc, qf, conv_h, dd,
                          nsub, fast=True, out_dir=out_dir)
        dt_s = time.perf_counter() - t0
        base = {'value': n * Tc * nsub / dt_s, 'unit': 'reach-steps/s', 'cores': 1, 'kind': 'port',
                'sample': f'{n} reaches x {Tc} runoff steps, direct-form convolution + unit_route, {dt_s:.2f} s, 1 thread'}
        chk = torch.zeros((Tc, n), dtype=torch.float64, device=dev)
        state.zero_(); q_ch.zero_(); q_full.zero_()
        plan.unit_route_uh_dev(q_ch, q_full, None, kern, state, n_ks, depth[:Tc].contiguous(), Tc, nsub, discharge=chk, stream=stream)
        torch.cuda.synchronize()
        got = chk.cpu().numpy()
        chk_kernel = plan.profile()['ticks_per_launch']
        if not np.allclose(got, dd, rtol=1e-10, atol=1e-10 * np.abs(dd).max()):
            raise SystemExit('bench.py: GPU UnitMuskingum result differs from the oracle; refusing to report a number')
        chk_name = {'direct': f'k_uh_convolve + k_direct<UNIT>, {chk_kernel} rows per task', 'tile': f'k_rec_in_uh + k_tile, {chk_kernel} ticks per task', 'tick': 'k_tick'}[plan.last_kernel()]
        base['parity_gate'] = (f'{Tc} rows x {n} reaches, convolution + routing by the timed kernels ({chk_name}) == oracle, '
                               f'rtol 1e-10, max |diff| {float(np.abs(got - dd).max()):.3e}')

    for _ in range(args.warmup):
        one_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    from river_route_amd.engine import copy_bandwidth
    from river_route_amd.measure import roofline_from_profile
    prof = plan.profile()
    aux = plan.profile_aux()
    # a year as the reference runs it: ten files of T steps, one call each, the discharge and the convolution's tail carried from
    # call to call (UnitMuskingum.py's loop over runoff files); the same depth block stands for each file
    year = None
    if T * 10 == 35_040 and not args.no_secondary:      # (--no-secondary: the counter passes sum every dispatch of the process)
        def year_pass():
            state.zero_(); q_ch.zero_(); q_full.zero_()
            for _ in range(10):
                plan.unit_route_uh_dev(q_ch, q_full, None, kern, state, n_ks, depth, T, nsub, discharge=out, stream=stream)
        plan.set_options(rows_per_chunk=args.chunk_rows, sample_every=0)
        year_pass()
        torch.cuda.synchronize()
        ty = time.perf_counter()
        year_pass()
        torch.cuda.synchronize()
        ty = time.perf_counter() - ty
        year = {'runoff_steps': 10 * T, 'calls': 10, 'ms': ty * 1e3, 'value': float(n) * 10 * T * nsub / ty, 'unit': 'reach-steps/s',
                'note': 'ten consecutive calls of the block above, discharge state and convolution tail carried between them'}
    if base is not None and prof['ticks_per_launch'] > 1 and chk_kernel <= 1:
        raise SystemExit('bench.py: the parity gate did not run the timed kernel; refusing to report a number')
    kern_name = plan.last_kernel()
    traffic = pmc_traffic(args.order, n, T, nsub, key='config4' if args.order == 'random' else 'config4_postorder') if n_ks == 48 else None
    roofline = roofline_from_profile(prof, nsub, HBM_PEAK_GBS, copy_gbs=copy_bandwidth(device_index), unit=True, kernel=kern_name, traffic=traffic)
    whole_path(roofline, prof, aux, kern_name, float(n) * T * nsub * args.steps / elapsed, traffic)
    if roofline is not None and 'k_rec_in' in roofline['path']['kernels']:
        roofline['path']['kernels'][f'k_rec_in_uh (convolution fused, {n_ks} taps)'] = roofline['path']['kernels'].pop('k_rec_in')
    how = ('the convolution as a pass of its own into work rows (k_uh_convolve_ring: not bracketed by the sampling events, see the kernel summary), then the direct row path'
           if kern_name == 'direct' else 'convolution fused into the record in-pass + routing')
    line = {'metric': 'reach-steps/sec', 'value': float(n) * T * nsub * args.steps / elapsed, 'unit': 'reach-steps/s',
            'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'UnitMuskingum, {n}-reach synthetic network + {n_ks}-step UH kernel, {T} runoff steps, '
                                   f'{nsub} sub-step(s), fp64, 1xMI355X (BASELINE config 4{", params in depth-first post-order" if args.order == "postorder" else ""}; {how})',
                       'reaches': n, 'runoff_steps': T, 'uh_steps': n_ks, 'headwaters': plan.n_headwaters, 'params_order': args.order, 'kernel': kern_name},
            'roofline': roofline, 'cpu_baseline': base}
    if year is not None:
        line['year'] = year
    plan.close()
    del depth, out, kern, state, q_ch, q_full
    torch.cuda.empty_cache()
    return line


def launch_ranks(n_ranks: int) -> int:
    """`python bench.py --gpus N` without torch.distributed.run: this process -- which never touches a GPU -- starts one
    fresh child per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (what torchrun would set), relays rank 0's
    JSON line and returns non-zero if any rank fails.  RR_DIST_BACKEND=gloo lets the ranks share one card (rehearsal).
    The children get HSA_ENABLE_IPC_MODE_LEGACY=0 in their environment (RCCL's device-memory sharing needs dmabuf IPC on these hosts and
    the variable is read when the HIP runtime starts)"""
    import socket
    import subprocess
    import threading
    if os.environ.get('RR_DIST_BACKEND', 'nccl') == 'nccl':
        import torch      # counts the devices (hipGetDeviceCount on this image: no context, no memory); this process runs nothing on them and
        have = torch.cuda.device_count()      # the ranks are fresh children, not forks or execs of it
        if have < n_ranks:
            print(f'bench.py --gpus {n_ranks}: {have} GPU(s) visible; RCCL needs one per rank '
                  f'(RR_DIST_BACKEND=gloo rehearses the same run with the ranks sharing a card)', file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    chunks = []      # rank 0's stdout, drained while it runs (a full pipe would stall it until the deadline)
    reader = threading.Thread(target=lambda: chunks.extend(iter(lambda: procs[0].stdout.read(65536), b'')), daemon=True)
    reader.start()
    deadline = time.monotonic() + float(os.environ.get('RR_BENCH_TIMEOUT', '3000'))
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = f'rank {r} exited with code {p.returncode}'
        if failed is None and time.monotonic() > deadline:
            failed = 'deadline (RR_BENCH_TIMEOUT) passed'
        if failed is None:
            time.sleep(0.2)
    if failed is None:
        for r, p in enumerate(procs):
            if p.returncode != 0:
                failed = f'rank {r} exited with code {p.returncode}'
    if failed is not None:
        for p in procs:      # exactly the children started above
            if p.poll() is None:
                p.kill()
    for p in procs:
        p.wait()
    reader.join(timeout=10)
    out = b''.join(chunks).decode('utf-8', 'replace')
    if failed is not None:
        sys.stderr.write(out)
        print(f'bench.py --gpus {n_ranks}: {failed}', file=sys.stderr)
        return 1
    lines = [ln for ln in out.splitlines() if ln.startswith('{')]
    if not lines:
        print(f'bench.py --gpus {n_ranks}: rank 0 printed no JSON line', file=sys.stderr)
        return 1
    print(lines[-1])
    return 0


def main():
    args = parse_args()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:      # stand-alone: be the launcher, before anything touches a GPU
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        args.gpus = world

    from river_route_amd import synth
    n, T, nsub, dt = args.reaches, args.runoff_steps, args.substeps, 900.0
    rows = min(args.forcing_rows or 288, T)
    net = indptr = indices = c1 = c2 = c3 = base = None
    if world == 1 and args.workload == 'rapid':
        net = synth.synth_network(n, order=args.order)
        indptr, indices = csc_from_down(net.down_index)
        c1, c2, c3 = muskingum_coefficients(net.k, net.x, dt)
        if not args.no_cpu_baseline:      # on the host cores, BEFORE this process touches the GPU (the replicas fork)
            base = cpu_baseline(net, indptr, indices, c1, c2, c3, dt, nsub, min(args.cpu_baseline_steps, rows),
                                args.cpu_baseline_seconds, args.cpu_replicas)

    import torch
    import torch.distributed as dist
    from river_route_amd import _lib
    from river_route_amd.engine import Plan, copy_bandwidth

    if not torch.cuda.is_available() or _lib.device_count() < 1:
        raise SystemExit('bench.py needs a GPU: the HIP engine has no CPU fallback')
    # one rank per GPU; RR_DIST_BACKEND=gloo lets several ranks share one card for a rehearsal on a 1-GPU box
    backend = os.environ.get('RR_DIST_BACKEND', 'nccl')
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    dev = torch.device('cuda', device_index)
    if world > 1:
        import datetime
        # a rank that never gets its message is reported by the exchange's own deadline (multi_gpu.run_distributed); this one
        # bounds the collectives around it
        limit = datetime.timedelta(seconds=float(os.environ.get('RR_DIST_TIMEOUT', '600')))
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev, timeout=limit)
        else:
            dist.init_process_group(backend, timeout=limit)
        from river_route_amd import multi_gpu
        multi_gpu.bench_main(args, rank, device_index, world, gate=None if (args.no_cpu_baseline or args.substeps != 1) else part_parity_gate)
        return
    local_rank = device_index

    if args.sequential_parts > 1:
        print(json.dumps(bench_sequential_parts(args, device_index)))
        return
    if args.workload == 'unit':
        print(json.dumps(bench_unit(args, device_index)))
        return
    if args.workload == 'rapid_f32':
        print(json.dumps(bench_rapid_f32(args, device_index)))
        return
    if args.workload == 'dropin':      # the host-array and file-to-file lines alone (they are `secondary` entries of the default run)
        for entry in bench_dropin(args, device_index):
            print(json.dumps(entry))
        return
    line = bench_rapid(args, device_index, net, indptr, indices, c1, c2, c3, base)
    if not args.no_secondary and n == 1_000_000 and T == 35_040 and nsub == 1:
        line['secondary'] = secondary_lines(args, device_index)
    print(json.dumps(line))


def part_parity_gate(eng, spec, coef, run, rows=96):
    """N > 1: this rank's part against the oracle before anything is timed.  The first `rows` forcing rows go through the
    distributed driver (`run`: the same exchange, engine and kernels as the timed passes) from a zero state into a plain
    array; the oracle routes the part's own reaches on the host, the discharge arriving over the cut -- the boundary series
    this rank RECEIVED, themselves rows the upstream ranks check against their own oracle runs -- folded into the lateral
    volume of the reach it enters (c2 Q[t-1] + c1 Q[t], _numba_kernels.py:70-78 for an upstream reach that is not in the
    local system).  One sub-step per row only.  Raises AssertionError on a mismatch."""
    import torch
    from oracle import oracle
    c1, c2, c3, c4_dt = coef
    ng, real = spec.n_ghost, spec.real_global
    Tg = int(min(rows, eng.T, eng.lat_rows))
    chk = torch.zeros((Tg, spec.n_local), dtype=torch.float64, device=eng.dev)
    eng.reshape_call(Tg, chk, Tg)
    run(Tg)
    torch.cuda.synchronize()
    got = chk.cpu().numpy()[:, ng:]
    G = eng.ghost_series[:Tg].cpu().numpy()
    E = eng.export_series[:Tg].cpu().numpy()
    c1r, c2r, c3r, c4r = c1[real], c2[real], c3[real], c4_dt[real]
    down = np.where(spec.down_local[ng:] >= 0, spec.down_local[ng:] - ng, -1)
    has = down >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
    indices = down[has].astype(np.int32)
    ql = eng.lateral[:Tg, ng:].cpu().numpy().copy()
    for g in range(ng):      # the engine's q0 is zero here, so a ghost's value before the first row is zero too
        d = int(spec.down_local[g] - ng)
        old = np.concatenate([[0.0], G[:-1, g]])
        ql[:, d] += (c2r[d] * old + c1r[d] * G[:, g]) / c4r[d]
    q, want = np.zeros(real.size), np.zeros((Tg, real.size))
    oracle.rapid_route(indptr, indices, -c1r[indices], c2r, c3r, c4r, q, ql, want, 1)
    scale = float(np.abs(want).max())
    assert np.allclose(got, want, rtol=1e-10, atol=1e-10 * scale), f'part {spec.part}: max |diff| {float(np.abs(got - want).max()):.3e} of {scale:.3e}'
    ex = np.searchsorted(real, spec.export_global)
    if ex.size:
        assert np.allclose(np.maximum(E[:, :ex.size], 0.0), want[:, ex], rtol=1e-10, atol=1e-10 * scale), f'part {spec.part}: export series differs'
    return (f'first {Tg} rows of its part through the timed exchange and kernels == oracle on the part with the received boundary '
            f'series folded in, rtol 1e-10')


def bench_sequential_parts(args, device_index, parts=None):
    """BASELINE config 5's network and partition on ONE card: `parts` x args.reaches reaches cut by rr_partition_forest exactly as
    `bench.py --gpus parts` cuts them, the parts routed one after another with their boundary series (the flow is one-directional:
    docs/references/parallelism.md:67-75; multi_gpu.run_sequential).  Gate: the first 96 rows of EVERY part against the oracle on the
    undivided network.  Timed: the year per part (warm-up + `steps` passes each); `value` = all reach-steps / the sum of the parts'
    times -- what one card does with the 10M-reach network -- and `config.part_*` what each GPU of the N-GPU run would have to do."""
    import torch
    from river_route_amd import synth
    from river_route_amd.engine import MODE_RAPID, partition_forest
    from river_route_amd.multi_gpu import HipPartEngine, run_sequential, split_network
    P = int(parts or args.sequential_parts)
    n, T, nsub, dt = args.reaches * P, args.runoff_steps, args.substeps, 900.0
    dev = torch.device('cuda', device_index)
    t0 = time.perf_counter()
    net = synth.synth_network(n, order=args.order)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = muskingum_coefficients(net.k, net.x, dt / nsub)
    c4_dt = (c1 + c2) / dt
    part_of, sizes = partition_forest(indptr, indices, P)
    specs = [split_network(net.down_index, part_of, p, P) for p in range(P)]
    setup_s = time.perf_counter() - t0
    gate = None
    if not args.no_cpu_baseline and nsub == 1:
        from oracle import oracle
        Tg = 96
        q_ref, d_ref = np.zeros(n), np.zeros((Tg, n))
        t0 = time.perf_counter()
        for r0 in range(0, Tg, 16):
            ql = synth.synth_qlateral_torch(n, r0, r0 + 16, dev, dt=dt).cpu().numpy()
            oracle.rapid_route(indptr, indices, -c1[indices], c2, c3, c4_dt, q_ref, ql, d_ref[r0:r0 + 16], 1)
            del ql
        oracle_s = time.perf_counter() - t0
        worst = [0.0]

        def check(spec, eng):
            torch.cuda.synchronize()
            got, want = eng.discharge.cpu().numpy()[:, spec.n_ghost:], d_ref[:, spec.real_global]
            scale = float(np.abs(want).max())
            if not (np.allclose(got, want, rtol=1e-10, atol=1e-10 * scale) and np.allclose(eng.final_state(), q_ref[spec.real_global], rtol=1e-10, atol=1e-10 * scale)):
                raise SystemExit(f'bench.py: part {spec.part} of the partitioned network differs from the oracle on the undivided network; refusing to report a number')
            worst[0] = max(worst[0], float(np.abs(got - want).max()) / scale)

        run_sequential(specs, lambda sp: HipPartEngine(sp, c1, c2, c3, c4_dt, np.zeros(n), synth.synth_qlateral_torch(n, 0, Tg, dev, columns=sp.real_global, dt=dt),
                                                       Tg, 1, device_index, out_rows=Tg), Tg, 1, check)
        gate = (f'{Tg} rows of every one of the {P} parts (boundary series handed from part to part) == oracle on the undivided {n}-reach network, '
                f'rtol 1e-10, worst difference {worst[0]:.2e} of the largest discharge; oracle {n * Tg / oracle_s:.3e} reach-steps/s on 1 host core')
        del d_ref
    rows = min(args.forcing_rows or 288, T)
    per_part = []

    def make(sp):
        lateral = synth.synth_qlateral_torch(n, 0, rows, dev, columns=sp.real_global, dt=dt)
        eng = HipPartEngine(sp, c1, c2, c3, c4_dt, np.zeros(n), lateral, T, nsub, device_index, out_rows=min(T, args.sink_rows or 128), sample_every=args.sample_every)
        eng.sched = eng.plan.reserve(MODE_RAPID, T, nsub)
        return eng

    def route(eng, T_, S_):
        ts = []
        for rep in range(args.warmup + args.steps):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            eng.begin(); eng.advance(T_, S_); eng.end()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t1)
        eng.pass_s = ts[args.warmup:]

    def visit(sp, eng):
        prof, tiles = eng.plan.profile(), eng.plan.tile_info()
        aux = eng.plan.profile_aux()
        launch_us = prof['sampled_ms'] / max(1, prof['brackets']) * 1e3
        per_part.append({'part': sp.part, 'reaches': int(sp.real_global.size), 'boundary_inflows': sp.n_ghost, 'exports': int(sp.export_global.size),
                         'feeds': [d for d, _ in sp.downstream_parts], 'fed_by': [u for u, _ in sp.upstream_parts], 'depth': eng.plan.depth,
                         'tiles': tiles['tiles'], 'tile_levels': tiles['levels'], 'kernel': eng.plan.last_kernel(), 'ticks_per_launch': eng.sched['ticks_per_launch'],
                         'ring_gb': round(eng.sched['ring_bytes'] / 1e9, 1), 'ms_per_year': [round(t * 1e3, 1) for t in eng.pass_s],
                         'routing_launch_us': round(launch_us, 1), 'routing_launches': prof['launches'],
                         'passes_us': {k: round(v['sampled_ms'] / max(1, v['sampled']) * 1e3, 1) for k, v in aux.items()}})

    run_sequential(specs, make, T, nsub, visit, route)
    total_s = sum(sum(p['ms_per_year']) for p in per_part) * 1e-3
    return {'metric': 'reach-steps/sec', 'value': float(n) * T * nsub * args.steps / total_s, 'unit': 'reach-steps/s', 'n_gpus': 1, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': total_s / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64',
            'data': 'synthetic',
            'config': {'workload': f'RapidMuskingum, ONE {n}-reach synthetic network (BASELINE config 5 at 8 x 1.25M) cut into {P} parts by rr_partition_forest, the parts '
                                   f'routed one after another on ONE MI355X with their boundary series, {T} runoff steps @ 900 s, {nsub} sub-step(s), fp64',
                       'reaches': n, 'parts': P, 'runoff_steps': T, 'substeps': nsub, 'params_order': args.order, 'setup_s': round(setup_s, 1), 'per_part': per_part,
                       'baseline_config': 5},
            'roofline': None, 'cpu_baseline': None if gate is None else {'parity_gate': gate}}


def secondary_lines(args, device_index):
    """BASELINE configs 2 and 4 in the driver's line, each behind its own oracle gate (bench_rapid / bench_unit refuse to
    return a number that does not reproduce the oracle): `value`, `ms_per_step`, `roofline`, a short `cpu_baseline`."""
    import copy
    from river_route_amd import synth
    out = []
    a = copy.copy(args)
    a.reaches, a.cpu_baseline_seconds, a.cpu_replicas = 100_000, 1.0, 0
    net = synth.synth_network(a.reaches, order=a.order)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = muskingum_coefficients(net.k, net.x, 900.0)
    base = None if a.no_cpu_baseline else cpu_baseline(net, indptr, indices, c1, c2, c3, 900.0, a.substeps, min(a.cpu_baseline_steps, a.forcing_rows or 288), 1.0, 0)
    line = bench_rapid(a, device_index, net, indptr, indices, c1, c2, c3, base)
    line['config']['baseline_config'] = 2
    out.append(line)
    a = copy.copy(args)
    a.workload = 'unit'
    line = bench_unit(a, device_index)
    line['config']['baseline_config'] = 4
    out.append(line)
    a = copy.copy(args)      # the same call on the post-order params table: UnitMuskingum on the direct row path
    a.workload, a.order, a.no_cpu_baseline = 'unit', 'postorder', args.no_cpu_baseline
    line = bench_unit(a, device_index)
    line['config']['variant_of_baseline_config'] = 4
    out.append(line)
    out.append(bench_rapid_f32(args, device_index))
    # the headline's network with its params file in depth-first post-order: the direct row path (no record ring and no permutation
    # pass for 95 % of the columns)
    a = copy.copy(args)
    a.order, a.cpu_baseline_seconds, a.cpu_replicas = 'postorder', 1.0, 0
    net = synth.synth_network(a.reaches, order=a.order)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = muskingum_coefficients(net.k, net.x, 900.0)
    base = None if a.no_cpu_baseline else cpu_baseline(net, indptr, indices, c1, c2, c3, 900.0, a.substeps, a.cpu_baseline_steps, 1.0, 0)
    line = bench_rapid(a, device_index, net, indptr, indices, c1, c2, c3, base)
    line['config']['variant_of_baseline_config'] = 3
    out.append(line)
    del net, indptr, indices, c1, c2, c3, base
    out.extend(bench_dropin(args, device_index))
    # BASELINE config 5's network (10M reaches, 8 parts) on this one card, the parts one after another, params in depth-first post-order:
    # every part on the direct row path with its boundary reaches (the random-order run of the same network: profiles/r05_parts_10M.txt)
    a = copy.copy(args)
    a.order, a.reaches, a.steps, a.warmup = 'postorder', 1_250_000, 1, 1
    out.append(bench_sequential_parts(a, device_index, parts=8))
    return out


def _rss_gb():
    import resource
    return round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 2)      # high-water mark of this process, GB


_ROUTE_CHILD = r'''
import json, resource, sys, time
import pandas as pd
try:
    import pyarrow  # noqa: F401
except ImportError:      # pandas without a parquet engine: the tables travel as pickles
    pd.read_parquet = lambda path, columns=None, **kw: (pd.read_pickle(path)[list(columns)] if columns is not None else pd.read_pickle(path))
    pd.DataFrame.to_parquet = lambda self, path, **kw: self.to_pickle(path)
import river_route_amd as rr
kind, kw = sys.argv[1], json.loads(sys.argv[2])
walls = []
import glob, os
for rep in range(2):      # the first pass is a cold process (library load, HIP context, page cache of the file); the second is what a run of many files sees
    for old in glob.glob(os.path.join(kw['discharge_dir'], '*')):      # (a run writes new files; truncating the first pass's would be timed otherwise)
        os.remove(old)
    t0 = time.perf_counter()
    r = getattr(rr, kind)(**kw)
    r.route()
    walls.append(time.perf_counter() - t0)
    kernel = r._plan.last_kernel()
    del r
hwm = [ln for ln in open('/proc/self/status') if ln.startswith('VmHWM')][0].split()      # (ru_maxrss starts at the parent's value in a forked child; VmHWM belongs to this program's own address space)
print(json.dumps({'walls': walls, 'kernel': kernel, 'peak_rss_gb': round(int(hwm[1]) * 1024 / 1e9, 2)}))
'''


def _route_in_child(kind, **kw):
    """<Router>(config).route() twice in a fresh process of its own, so that wall time and peak host RSS are the router's, not this bench's
    (which holds the synthetic input arrays).  The child is started, not exec'ed: this process keeps its GPU context."""
    import subprocess
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.dirname(os.path.abspath(__file__))] + ([os.environ['PYTHONPATH']] if os.environ.get('PYTHONPATH') else [])))
    res = subprocess.run([sys.executable, '-c', _ROUTE_CHILD, kind, json.dumps(kw)], capture_output=True, text=True, env=env)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    if res.returncode != 0 or not lines:
        raise SystemExit(f'bench.py: {kind}(config).route() failed in its child process:\n{res.stderr[-2000:]}')
    return json.loads(lines[-1])


def bench_dropin(args, device_index, T: int = 744):
    """What a drop-in caller gets, PCIe and files included (the shape of the reference's own harness, tests/test_zbenchmarks.py:29-152:
    wall time and peak memory of route-from-qlateral, route-from-depths, end to end through .route()), each checked against the oracle
    on its first rows: (i) kernels.rapid_route with numpy arrays -- the three-line swap of INTEGRATION.md, host pointers through the
    C ABI; (ii) RapidMuskingum(config).route() from a float32 qlateral netCDF to a discharge netCDF; (iii) UnitMuskingum(config)
    .route() from runoff depths.  One month of hourly rows (744) on the headline's network.  `value` is reach-steps per second of wall time."""
    import shutil
    import tempfile
    import pandas as pd
    from scipy.io import netcdf_file
    import scipy.sparse
    from oracle import oracle
    import river_route_amd as rr
    from river_route_amd import kernels, synth
    try:
        import pyarrow  # noqa: F401
    except ImportError:      # this image has pandas without a parquet engine: params / state tables travel as pickles (as in tests/conftest.py)
        pd.read_parquet = lambda path, columns=None, **kw: (pd.read_pickle(path)[list(columns)] if columns is not None else pd.read_pickle(path))
        pd.DataFrame.to_parquet = lambda self, path, **kw: self.to_pickle(path)
    n, dt = args.reaches, 3600.0
    net = synth.synth_network(n, order=args.order)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = muskingum_coefficients(net.k, net.x, dt)
    lhs, c4_dt = -c1[indices], (c1 + c2) / dt
    ql = synth.synth_qlateral(n, 0, T, dt=dt)
    chk = 48
    q_ref, d_ref = np.zeros(n), np.zeros((chk, n))
    oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, ql[:chk], d_ref, 1)
    scale = float(np.abs(d_ref).max())
    out = []

    def entry(what, seconds, gate, extra=None):
        e = {'metric': 'reach-steps/sec', 'value': float(n) * T / seconds, 'unit': 'reach-steps/s', 'n_gpus': 1, 'steps': 1, 'warmup': 1,
             'ms_per_step': seconds * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
             'config': {'workload': what, 'reaches': n, 'runoff_steps': T, 'params_order': args.order, 'timed_region': 'wall clock of the call, host arrays / files in and out',
                        'bench_process_peak_rss_gb': _rss_gb()},
             'roofline': None, 'cpu_baseline': {'parity_gate': gate}}
        if extra:
            e['config'].update(extra)
        out.append(e)

    # (i) the kernel boundary with numpy arrays
    d = np.zeros((T, n))
    for rep in range(2):      # the first call builds and caches the plan, as the reference's first call compiles
        q = np.zeros(n)
        t0 = time.perf_counter()
        kernels.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q, ql, d, 1)
        sec = time.perf_counter() - t0
    if not np.allclose(d[:chk], d_ref, rtol=1e-10, atol=1e-10 * scale):
        raise SystemExit('bench.py: kernels.rapid_route differs from the oracle; refusing to report a number')
    entry(f'kernels.rapid_route(numpy arrays): {n} reaches x {T} hourly rows, float64 host arrays in and out through rr_rapid_route (PCIe both ways)', sec,
          f'first {chk} rows == oracle, rtol 1e-10')
    kernels.clear_plan_cache()
    del d

    tmp = tempfile.mkdtemp(prefix='rr_bench_')
    try:
        params = os.path.join(tmp, 'params.parquet')
        pd.DataFrame({'river_id': net.river_ids, 'downstream_river_id': net.downstream_ids, 'k': net.k, 'x': net.x}).to_parquet(params)
        dates = (np.datetime64('2020-01-01T00:00:00', 's') + np.arange(T) * np.timedelta64(int(dt), 's')).astype('datetime64[s]').astype(np.int64).astype(np.float64)

        def write_rows(path, var, rows32):
            with netcdf_file(path, 'w', version=2) as ds:
                ds.createDimension('time', None)      # record dimension: a fixed NetCDF-3 variable holds less than 2 GiB
                ds.createDimension('river_id', n)
                tv = ds.createVariable('time', 'f8', ('time',))
                tv.units = 'seconds since 1970-01-01 00:00:00'
                tv[:] = dates
                rid = ds.createVariable('river_id', 'i4', ('river_id',))
                rid[:] = net.river_ids.astype(np.int32)
                v = ds.createVariable(var, 'f4', ('time', 'river_id'))
                v[:] = rows32

        def read_q(path, rows):
            with netcdf_file(path, 'r', mmap=True) as ds:
                return np.array(ds.variables['Q'][:rows], dtype=np.float32)

        # (ii) RapidMuskingum file to file, float32 lateral volumes as qlateral files store them
        ql32 = ql.astype(np.float32)
        qfile = os.path.join(tmp, 'qlateral.nc')
        write_rows(qfile, 'qlateral', ql32)
        q_ref32, d_ref32 = np.zeros(n), np.zeros((chk, n))
        oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref32, ql32[:chk].astype(np.float64), d_ref32, 1)
        del ql32
        os.makedirs(os.path.join(tmp, 'rapid'))
        child = _route_in_child('RapidMuskingum', params_file=params, qlateral_files=[qfile], discharge_dir=os.path.join(tmp, 'rapid'), dt_routing=int(dt), log=False)
        sec = child['walls'][1]
        got = read_q(os.path.join(tmp, 'rapid', 'discharge_qlateral.nc'), chk)
        if not np.allclose(got, d_ref32.astype(np.float32), rtol=1.2e-7, atol=1e-10 * scale):
            raise SystemExit('bench.py: RapidMuskingum(config).route() differs from the oracle; refusing to report a number')
        entry(f'RapidMuskingum(config).route(): {n} reaches x {T} hourly rows, float32 qlateral netCDF in, float32 discharge netCDF out (params parquet read, '
              f'network analysis, file read, upload, routing, download, file write)', sec, f'first {chk} rows of the discharge file == oracle, <= 1 ulp(float32)',
              {'peak_host_rss_gb': child['peak_rss_gb'], 'cold_process_wall_s': round(child['walls'][0], 3), 'routing_kernel': child['kernel'],
               'measured_in': 'a fresh child process (the second of two route() calls is `value`; the first, cold, one is cold_process_wall_s)'})
        os.remove(qfile)
        shutil.rmtree(os.path.join(tmp, 'rapid'))

        # (iii) UnitMuskingum from runoff depths + a 48-step unit-hydrograph kernel
        n_ks = args.uh_steps
        kern = synth.synth_uh_kernel(n, n_ks, tr=dt)
        kfile = os.path.join(tmp, 'uh_kernel.npz')
        scipy.sparse.save_npz(kfile, scipy.sparse.csr_matrix(kern), compressed=False)
        depth32 = synth.synth_runoff_depth(n, 0, T).astype(np.float32)
        dfile = os.path.join(tmp, 'depth.nc')
        write_rows(dfile, 'qlateral', depth32)
        from tests_support import unit_split_arrays
        hw_idx, inner_idx, A_in, A_hw = unit_split_arrays(indptr, indices, n)
        c1i, c2i, c3i = c1[inner_idx], c2[inner_idx], c3[inner_idx]
        conv = oracle.UnitHydrograph(kern).convolve(depth32[:chk].astype(np.float64))
        qc, qf, dd = np.zeros(inner_idx.size), np.zeros(inner_idx.size), np.zeros((chk, n))
        oracle.unit_route(A_in.indptr, A_in.indices, -c1i[A_in.indices], A_in.indptr, A_in.indices, A_in.data, A_hw.indptr, A_hw.indices, A_hw.data,
                          c1i, c2i, c3i, hw_idx, inner_idx, qc, qf, conv, dd, 1)
        del depth32, kern, conv
        os.makedirs(os.path.join(tmp, 'unit'))
        child = _route_in_child('UnitMuskingum', params_file=params, qlateral_files=[dfile], discharge_dir=os.path.join(tmp, 'unit'), uh_kernel_file=kfile, dt_routing=int(dt), log=False)
        sec = child['walls'][1]
        got = read_q(os.path.join(tmp, 'unit', 'discharge_depth.nc'), chk)
        if not np.allclose(got, dd.astype(np.float32), rtol=1.2e-7, atol=1e-9 * float(np.abs(dd).max())):
            raise SystemExit('bench.py: UnitMuskingum(config).route() differs from the oracle; refusing to report a number')
        entry(f'UnitMuskingum(config).route(): {n} reaches x {T} hourly rows of float32 runoff depths + {n_ks}-step unit-hydrograph kernel (npz), float32 discharge '
              f'netCDF out', sec, f'first {chk} rows of the discharge file == oracle (direct-form convolution + unit_route), <= 1 ulp(float32)',
              {'uh_steps': n_ks, 'peak_host_rss_gb': child['peak_rss_gb'], 'cold_process_wall_s': round(child['walls'][0], 3), 'routing_kernel': child['kernel'],
               'measured_in': 'a fresh child process (the second of two route() calls is `value`; the first, cold, one is cold_process_wall_s)'})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def bench_rapid_f32(args, device_index, factor: int = 4):
    """The headline's network and year the way the routers run a float32 qlateral file with hourly output: float32 lateral
    rows in (rr_rapid_route_f32in_dev, exact in float64), float32 rows out, each the mean of `factor` routed rows
    (TransformMuskingum.py:128-142 fused into the out-pass): 4 + 8 B through the in-pass and 8 + 4 / factor B through the
    out-pass per reach-step instead of 8 + 8 and 8 + 8.  Gate: the first 96 rows against the oracle on the float64 copy of
    the same float32 values, mean and cast as numpy does them."""
    import torch
    from river_route_amd import synth
    from river_route_amd.engine import Plan
    from river_route_amd.measure import roofline_from_profile
    n, T, nsub, dt = args.reaches, args.runoff_steps, 1, 900.0
    rows = min(args.forcing_rows or 288, T)
    dev = torch.device('cuda', device_index)
    net = synth.synth_network(n, order=args.order)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = muskingum_coefficients(net.k, net.x, dt)
    plan = Plan(indptr, indices, device=device_index)
    plan.set_coeffs(-c1[indices], c2, c3, (c1 + c2) / dt)
    plan.set_options(rows_per_chunk=args.chunk_rows, sample_every=args.sample_every)
    ql32 = synth.synth_qlateral_torch(n, 0, rows, dev, dt=dt).to(torch.float32)
    out32 = torch.zeros((T // factor, n), dtype=torch.float32, device=dev)
    q_t = torch.zeros(n, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    gate = None
    if not args.no_cpu_baseline:
        from oracle import oracle
        chk_T = 96
        ql_h = ql32[:chk_T].cpu().numpy().astype(np.float64)
        q_ref, d_ref = np.zeros(n), np.zeros((chk_T, n))
        oracle.rapid_route(indptr, indices, -c1[indices], c2, c3, (c1 + c2) / dt, q_ref, ql_h, d_ref, 1)
        want = d_ref.reshape(chk_T // factor, factor, n).mean(axis=1).astype(np.float32)
        chk = torch.zeros((chk_T // factor, n), dtype=torch.float32, device=dev)
        plan.rapid_route_f32in_dev(q_t, ql32, rows, chk_T, 1, discharge32=chk, factor=factor, stream=stream)
        torch.cuda.synchronize()
        got = chk.cpu().numpy()
        if not np.allclose(got, want, rtol=1.2e-7, atol=1e-10 * float(np.abs(want).max())):
            raise SystemExit('bench.py: float32 path differs from the oracle; refusing to report a number')
        gate = f'{chk_T} rows x {n} reaches == oracle on the float64 copy, mean of {factor} rows and float32 cast as numpy, <= 1 ulp(f32)'

    def one_pass():
        q_t.zero_()
        plan.rapid_route_f32in_dev(q_t, ql32, rows, T, 1, discharge32=out32, factor=factor, stream=stream)

    for _ in range(args.warmup):
        one_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    traffic = pmc_traffic(args.order, n, T, 1, key='f32') if factor == 4 else None
    roofline = roofline_from_profile(plan.profile(), 1, HBM_PEAK_GBS, traffic=traffic)
    whole_path(roofline, plan.profile(), plan.profile_aux(), plan.last_kernel(), float(n) * T * args.steps / elapsed, traffic)
    plan.close()
    return {'metric': 'reach-steps/sec', 'value': float(n) * T * args.steps / elapsed, 'unit': 'reach-steps/s', 'n_gpus': 1, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'RapidMuskingum, {n}-reach synthetic network, {T} runoff steps @ 900 s, fp64 arithmetic, float32 lateral rows in '
                                   f'({rows}-row cyclic array) and float32 discharge out as means of {factor} rows ({T // factor} rows, all kept), 1xMI355X',
                       'reaches': n, 'runoff_steps': T, 'rows_per_output': factor, 'params_order': args.order, 'variant_of_baseline_config': 3},
            'roofline': roofline, 'cpu_baseline': None if gate is None else {'parity_gate': gate}}


def bench_rapid(args, local_rank, net, indptr, indices, c1, c2, c3, base):
    """RapidMuskingum, one year per bench step on one GPU (BASELINE config 3 at 1M reaches, config 2 at 100k): the line."""
    import torch
    from river_route_amd import synth
    from river_route_amd.engine import Plan, copy_bandwidth
    n, T, nsub, dt = net.n, args.runoff_steps, args.substeps, 900.0
    dev = torch.device('cuda', local_rank)
    plan = Plan(indptr, indices, device=local_rank)
    plan.set_coeffs(-c1[indices], c2, c3, (c1 + c2) / (dt * nsub))
    plan.set_options(rows_per_chunk=args.chunk_rows, sample_every=args.sample_every)
    # The direct row path (post-order params files) reads and writes the rows from its routing kernel: a tile walks down its own
    # columns, so the cyclic arrays must be longer than the caches hold for the whole chip (1,024 rows = 8 GB at 1M reaches).
    direct = plan.reserve(0, T, nsub)['direct']
    rows = min(args.forcing_rows or (1024 if direct else 288), T)

    ql = synth.synth_qlateral_torch(n, 0, rows, dev, dt=dt * nsub)      # the same bits as synth_qlateral (tests/test_host.py), made on the device
    sink_rows = min(T, args.sink_rows or (1024 if direct else plan.tile_info()['batch_rows']))      # record path: rows of one out-pass batch, a launch never writes a sink row twice
    out = torch.zeros((sink_rows, n), dtype=torch.float64, device=dev)
    q_t = torch.zeros(n, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def one_pass():
        q_t.zero_()
        plan.rapid_route_dev(q_t, ql, rows, out, sink_rows, T, nsub, stream)

    # Parity gate before timing, through the kernels the timed passes run (the time-tiled kernel and the record
    # permutation passes take every call of 32 sub-steps or more; on a post-order network the direct row path does): the first
    # rows of the forcing routed from a zero state into a plain array, compared element by element with the oracle's first pass.
    chk_name = None
    if base is not None:
        want = base.pop('first_pass')
        chk_T = want.shape[0]
        chk_out = torch.zeros((chk_T, n), dtype=torch.float64, device=dev)
        q_t.zero_()
        plan.rapid_route_dev(q_t, ql, rows, chk_out, chk_T, chk_T, nsub, stream)
        torch.cuda.synchronize()
        chk_kernel, chk_name = plan.profile()['ticks_per_launch'], plan.last_kernel()
        got = chk_out.cpu().numpy()
        if not np.allclose(got, want, rtol=1e-10, atol=1e-10 * np.abs(want).max()):
            raise SystemExit('bench.py: GPU result differs from the oracle; refusing to report a number')
        base['parity_gate'] = (f'{chk_T} rows x {n} reaches routed by the timed kernel family '
                               f'({KERNEL_NAMES[chk_name] + ", " + str(chk_kernel) + " ticks per task" if chk_kernel > 1 else "k_tick"}) == oracle, '
                               f'rtol 1e-10, max |diff| {float(np.abs(got - want).max()):.3e}')
        del chk_out

    for _ in range(args.warmup):
        one_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    prof, aux, kern = plan.profile(), plan.profile_aux(), plan.last_kernel()      # HIP events of the last timed pass, on the engine's stream
    if base is not None and chk_name != kern:
        raise SystemExit('bench.py: the parity gate did not run the timed kernel; refusing to report a number')
    reach_steps = float(n) * T * nsub * args.steps
    from river_route_amd.measure import roofline_from_profile
    copy_gbs = copy_bandwidth(local_rank)
    traffic = pmc_traffic(args.order, n, T, nsub, key='config2' if n == 100_000 and args.order == 'random' else None)
    roofline = roofline_from_profile(prof, nsub, HBM_PEAK_GBS, copy_gbs=copy_gbs, kernel=kern, traffic=traffic)
    whole_path(roofline, prof, aux, kern, reach_steps / elapsed, traffic)
    tiles, dinfo = plan.tile_info(), plan.direct_info()
    line = {
        'metric': 'reach-steps/sec', 'value': reach_steps / elapsed, 'unit': 'reach-steps/s',
        'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': f'RapidMuskingum, {n}-reach synthetic random-topology network '
                               f'(depth {plan.depth}, {"depth-first post-order" if args.order == "postorder" else args.order + " topological order"}), {T} runoff steps @ 900 s '
                               f'(1 yr @ 15 min), {nsub} sub-step(s), fp64, 1xMI355X',
                   'reaches': n, 'runoff_steps': T, 'substeps': nsub, 'network_depth': plan.depth,
                   'forcing': f'{rows}-row device-resident cyclic array', 'discharge_sink': f'{sink_rows}-row device-resident cyclic array', 'params_order': args.order,
                   'routing_kernel': KERNEL_NAMES[kern],
                   'permutation_passes_in_timed_region': kern != 'direct'},
        'roofline': roofline,
        'cpu_baseline': base,
    }
    if kern == 'direct':
        line['config'].update(direct_tiles=dinfo['tiles'], holes=dinfo['holes'], skeleton_positions=dinfo['skeleton_positions'],
                              skeleton_tile_levels=dinfo['skeleton_levels'], window_rows=dinfo['window_rows'])
    else:
        line['config'].update(tiles=tiles['tiles'], tile_levels=tiles['levels'], ghost_positions=tiles['ghosts'])
    plan.close()
    del ql, out, q_t
    torch.cuda.empty_cache()
    return line


from river_route_amd.measure import engine_sha16, pmc_traffic, whole_path      # noqa: E402  (HIP-event profile + counter passes -> the `roofline` object)

KERNEL_NAMES = {'tick': 'k_tick (streaming, one launch per tick)', 'tile': 'k_tile (time-tiled over subtree tiles and records) + k_rec_in / k_rec_out',
                'direct': 'k_direct (column-range tiles reading and writing the rows) + k_tile on the skeleton + k_rec_out over the holes'}


if __name__ == '__main__':
    main()

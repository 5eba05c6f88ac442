"""
bench.py -- reach-steps/sec of the Muskingum routing hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W

Workload at N=1 (BASELINE.json configs[2]): RapidMuskingum on the 1M-reach synthetic network, 1 year at
15-minute steps (35,040 runoff steps, dt_routing = dt_runoff = 900 s, fp64).  One bench "step" is ONE pass of
the hot path over that year: a single rr_rapid_route_dev call.  A year of lateral inflow for 1M reaches is
280 GB, so the forcing is a 96-row (one day) device-resident array read cyclically (row t % 96) and the
discharge goes to a 96-row cyclic sink; every routed row is still read from and written to HBM, and the
params-order <-> engine-order permutation passes are inside the timed region.  Inputs are resident in HBM
when the timed region starts (the PCIe-inclusive host-pointer rate is noted in DESIGN.md, never here).

The JSON line carries `roofline` for the dominant kernel (the routing tick, k_tick) from HIP events recorded
by the engine around sampled launches on its own stream, and `cpu_baseline`: the oracle (oracle/rr_oracle.c,
-O3 -march=native -ffast-math, 1 thread -- the reference path is single-threaded by construction) timed on a
bounded sample of the same workload on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
BYTES_PER_REACH_SUBSTEP = 72    # structure 8 + coefficients 32 + state 32
BYTES_PER_REACH_ROW = 16        # lateral read 8 + discharge write 8


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--reaches', type=int, default=1_000_000, help='reaches per GPU')
    ap.add_argument('--runoff-steps', type=int, default=35_040, help='runoff steps per bench step (1 yr @ 15 min)')
    ap.add_argument('--substeps', type=int, default=1)
    ap.add_argument('--forcing-rows', type=int, default=96)
    ap.add_argument('--order', default='random', choices=['random', 'levels', 'bfs'])
    ap.add_argument('--sample-every', type=int, default=128)
    ap.add_argument('--chunk-rows', type=int, default=16)
    ap.add_argument('--cpu-baseline-steps', type=int, default=96)
    ap.add_argument('--cpu-baseline-seconds', type=float, default=12.0)
    ap.add_argument('--exchange-rows', type=int, default=128,
                    help='N>1: runoff steps per boundary-series message')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--traffic-bytes-per-launch', type=float, default=None,
                    help='HBM bytes per routing-tick launch from a separate rocprofv3 --pmc run (profiles/)')
    return ap.parse_args()


def csc_from_down(down_index):
    has = down_index >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
    return indptr, down_index[has].astype(np.int32)


def muskingum_coefficients(k, x, dt):
    """river_route/routers/Muskingum.py:174-179."""
    r = dt / k
    den = r + 2.0 * (1.0 - x)
    return (r - 2.0 * x) / den, (r + 2.0 * x) / den, (2.0 * (1.0 - x) - r) / den


def cpu_baseline(net, indptr, indices, c1, c2, c3, dt, nsub, steps, seconds):
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
    check_row = None
    reps, dt_s = 0, 0.0
    while dt_s < seconds and reps < 64:      # the same `steps` rows again and again, state carried over
        t0 = time.perf_counter()
        oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q, ql, d, nsub, fast=True, out_dir=out_dir)
        dt_s += time.perf_counter() - t0
        reps += 1
        if check_row is None:
            check_row = d[min(steps, 8) - 1].copy()
    return {'value': n * steps * nsub * reps / dt_s, 'unit': 'reach-steps/s', 'cores': 1, 'kind': 'port',
            'sample': f'{n} reaches x {steps * reps} runoff steps ({reps} passes over {steps} forcing rows) x {nsub} '
                      f'sub-step(s), {dt_s:.2f} s, oracle/rr_oracle.c gcc -O3 -march=native -ffast-math, 1 thread '
                      f'of {os.cpu_count()} host cores',
            'check_row': check_row}


def main():
    args = parse_args()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)')
        args.gpus = world

    import torch
    import torch.distributed as dist
    from river_route_amd import _lib, synth
    from river_route_amd.engine import Plan

    if not torch.cuda.is_available() or _lib.device_count() < 1:
        raise SystemExit('bench.py needs a GPU: the HIP engine has no CPU fallback')
    # one rank per GPU; RR_DIST_BACKEND=gloo lets several ranks share one card for a rehearsal on a 1-GPU box
    backend = os.environ.get('RR_DIST_BACKEND', 'nccl')
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    dev = torch.device('cuda', device_index)
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
        from river_route_amd import multi_gpu
        multi_gpu.bench_main(args, rank, device_index, world)
        return
    local_rank = device_index

    n, T, nsub, dt = args.reaches, args.runoff_steps, args.substeps, 900.0
    net = synth.synth_network(n, order=args.order)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = muskingum_coefficients(net.k, net.x, dt)
    plan = Plan(indptr, indices, device=local_rank)
    plan.set_coeffs(-c1[indices], c2, c3, (c1 + c2) / (dt * nsub))
    plan.set_options(rows_per_chunk=args.chunk_rows, sample_every=args.sample_every)

    rows = min(args.forcing_rows, T)
    ql = torch.from_numpy(synth.synth_qlateral(n, 0, rows, dt=dt * nsub)).to(dev)
    out = torch.zeros((rows, n), dtype=torch.float64, device=dev)
    q_t = torch.zeros(n, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def one_pass():
        q_t.zero_()
        plan.rapid_route_dev(q_t, ql, rows, out, rows, T, nsub, stream)

    # parity spot check before timing: first rows of a short pass against the oracle
    base = None
    if not args.no_cpu_baseline:
        base = cpu_baseline(net, indptr, indices, c1, c2, c3, dt, nsub, min(args.cpu_baseline_steps, rows),
                            args.cpu_baseline_seconds)
        chk_T = min(rows, 8)
        chk_out = torch.zeros((chk_T, n), dtype=torch.float64, device=dev)
        q_t.zero_()
        plan.rapid_route_dev(q_t, ql, rows, chk_out, chk_T, chk_T, nsub, stream)
        torch.cuda.synchronize()
        got = chk_out[chk_T - 1].cpu().numpy()
        want = base.pop('check_row')
        if not np.allclose(got, want, rtol=1e-10, atol=1e-10 * np.abs(want).max()):
            raise SystemExit('bench.py: GPU result differs from the oracle; refusing to report a number')

    for _ in range(args.warmup):
        one_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    prof = plan.profile()      # HIP events of the last timed pass, on the engine's stream
    reach_steps = float(n) * T * nsub * args.steps
    bytes_per_reach_tick = BYTES_PER_REACH_SUBSTEP + BYTES_PER_REACH_ROW / nsub
    roofline = None
    if prof['sampled'] > 0 and prof['sampled_ms'] > 0:
        avg_ms = prof['sampled_ms'] / prof['sampled']
        avg_reaches = prof['sampled_reaches'] / prof['sampled']
        achieved = bytes_per_reach_tick * avg_reaches / (avg_ms * 1e-3) / 1e9
        roofline = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(achieved / HBM_PEAK_GBS, 4),
                    'traffic': args.traffic_bytes_per_launch,
                    'kernel': 'k_tick<lateral, 1 sub-step>' if nsub == 1 else 'k_tick<lateral, sub-steps>',
                    'avg_launch_us': round(avg_ms * 1e3, 3), 'min_launch_us': round(prof['min_ms'] * 1e3, 3),
                    'max_launch_us': round(prof['max_ms'] * 1e3, 3),
                    'algorithmic_bytes_per_launch': round(bytes_per_reach_tick * avg_reaches),
                    'launches_per_pass': prof['launches'], 'launches_sampled': prof['sampled'],
                    'pass_region_ms': round(prof['region_ms'], 3)}
    line = {
        'metric': 'reach-steps/sec', 'value': reach_steps / elapsed, 'unit': 'reach-steps/s',
        'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': f'RapidMuskingum, {n}-reach synthetic random-topology network '
                               f'(depth {plan.depth}, {args.order} topological order), {T} runoff steps @ 900 s '
                               f'(1 yr @ 15 min), {nsub} sub-step(s), fp64, 1xMI355X',
                   'reaches': n, 'runoff_steps': T, 'substeps': nsub, 'network_depth': plan.depth,
                   'forcing': f'{rows}-row device-resident cyclic array', 'params_order': args.order,
                   'permutation_passes_in_timed_region': not plan.identity_order},
        'roofline': roofline,
        'cpu_baseline': base,
    }
    print(json.dumps(line))


if __name__ == '__main__':
    main()

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
    ap.add_argument('--workload', default='rapid', choices=['rapid', 'unit'],
                    help="'unit' = BASELINE config 4 (UnitMuskingum + 48-step UH kernel); secondary line, not the headline")
    ap.add_argument('--uh-steps', type=int, default=48)
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


def bench_unit(args, device_index):
    """BASELINE config 4: UnitMuskingum, 1M reaches, 48-step UH kernel.  One bench step = convolve + route one
    block of runoff depths (default 3,504 steps = 1/10 year; the (T, n) depth block must be resident for the
    convolution, 28 GB at 1M reaches)."""
    import torch
    from river_route_amd import synth
    from river_route_amd.engine import Plan, uh_convolve_dev
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
    conv = torch.empty_like(depth)
    out_rows = 96
    out = torch.zeros((out_rows, n), dtype=torch.float64, device=dev)
    n_inner = plan.n_inner
    q_ch = torch.zeros(n_inner, dtype=torch.float64, device=dev)
    q_full = torch.zeros(n_inner, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def one_pass():
        state.zero_(); q_ch.zero_(); q_full.zero_()
        uh_convolve_dev(kern, state, depth, conv, T, n_ks, n, device=device_index, stream=stream)
        plan.unit_route_dev(q_ch, q_full, conv, T, out, out_rows, T, nsub, stream)

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
        conv_c = torch.empty((Tc, n), dtype=torch.float64, device=dev)
        uh_convolve_dev(kern, state, depth[:Tc].contiguous(), conv_c, Tc, n_ks, n, device=device_index, stream=stream)
        plan.unit_route_dev(q_ch, q_full, conv_c, Tc, chk, Tc, Tc, nsub, stream)
        torch.cuda.synchronize()
        got = chk.cpu().numpy()
        if not np.allclose(got, dd, rtol=1e-10, atol=1e-10 * np.abs(dd).max()):
            raise SystemExit('bench.py: GPU UnitMuskingum result differs from the oracle; refusing to report a number')

    for _ in range(args.warmup):
        one_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    line = {'metric': 'reach-steps/sec', 'value': float(n) * T * nsub * args.steps / elapsed, 'unit': 'reach-steps/s',
            'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'UnitMuskingum, {n}-reach synthetic network + {n_ks}-step UH kernel, {T} runoff steps, '
                                   f'{nsub} sub-step(s), fp64, 1xMI355X (BASELINE config 4; convolution + routing)',
                       'reaches': n, 'runoff_steps': T, 'uh_steps': n_ks, 'headwaters': plan.n_headwaters},
            'roofline': None, 'cpu_baseline': base}
    print(json.dumps(line))


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

    if args.workload == 'unit':
        bench_unit(args, device_index)
        return
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
    # roofline of the dominant kernel: the routing kernel (k_wave: one launch = `ticks_per_launch` routing ticks over
    # all reaches).  `achieved` prices the launch at the ALGORITHMIC bytes of SURVEY section 8(d) (streaming model,
    # 88 B per reach-step at nsub=1), which a time-tiled kernel legitimately undercuts; `traffic` is what it
    # really moved (separate rocprofv3 --pmc passes, profiles/r01_pmc_traffic.json; default shape at 1M reaches).
    traffic = args.traffic_bytes_per_launch
    traffic_per_reach_tick = None
    kernel_name = None
    if traffic is None and n == 1_000_000 and nsub == 1 and not plan.identity_order and \
            not any(k.startswith(('RR_WAVE', 'RR_REC')) for k in os.environ):
        try:
            with open(os.path.join(REPO, 'profiles', 'r01_pmc_traffic.json')) as f:
                traffic_per_reach_tick = json.load(f)['kernels']['k_wave_rec']['hbm_bytes_per_reach_tick']
            kernel_name = 'k_wave_rec (time-tiled routing over tick-indexed records)'
        except (OSError, KeyError, ValueError):
            traffic_per_reach_tick = None
    from river_route_amd.multi_gpu import roofline_from_profile
    roofline = roofline_from_profile(prof, nsub, traffic, HBM_PEAK_GBS, traffic_per_reach_tick)
    if roofline and kernel_name:
        roofline['kernel'] = kernel_name
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

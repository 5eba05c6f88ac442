"""
The partitioned path: forest partitioning, local networks with ghost/export reaches, and the batched one-way
boundary exchange.  CPU: world_size-2 gloo run with an oracle-backed engine injected (tests/multi_helpers.py)
against the single-domain oracle.  GPU: the real engine, all parts on one card through the in-process runner
(the same driver and the same C-ABI streaming calls the RCCL run uses), against the single-plan result.
"""
import os
import socket

import numpy as np
import pytest

from conftest import assert_close
from multi_helpers import OraclePartEngine, gloo_worker, hip_worker, setup_case
from river_route_amd import synth
from river_route_amd.engine import partition_forest
from river_route_amd.multi_gpu import run_in_process, split_network


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('n,parts', [(9, 2), (1000, 2), (1000, 8), (50001, 5)])
def test_partition_and_split_invariants(n, parts):
    net, indptr, indices, *_ = setup_case(n)
    part_of, sizes = partition_forest(indptr, indices, parts)
    assert sizes.sum() == n and np.array_equal(np.bincount(part_of, minlength=parts), sizes)
    has = net.down_index >= 0
    # parts are numbered upstream-first: discharge never flows to a lower-numbered part
    assert np.all(part_of[has] <= part_of[net.down_index[has]])
    if n >= 1000:
        assert sizes.max() <= 1.35 * n / parts
    seen = np.zeros(n, dtype=int)
    n_ghost = n_export = 0
    for p in range(parts):
        spec = split_network(net.down_index, part_of, p, parts)
        seen[spec.real_global] += 1
        n_ghost += spec.n_ghost
        n_export += spec.export_global.size
        # local order is upstream -> downstream and ghosts are headwaters
        loc = spec.down_local
        assert np.all((loc < 0) | (loc > np.arange(loc.size)))
        assert not np.isin(np.arange(spec.n_ghost), loc).any()
        assert np.all(part_of[spec.ghost_global] == spec.ghost_owner) and np.all(spec.ghost_owner < p)
        assert np.all(spec.export_consumer > p)
    assert np.all(seen == 1) and n_ghost == n_export == int((has & (part_of != part_of[np.maximum(net.down_index, 0)])).sum())


def test_partition_shape_keeps_the_pipeline_shallow():
    """The trunk partition: every cut edge enters the last part, parts are even; a chain, which
    has no trunk to speak of, falls back to the nested min-max cut and still satisfies the invariants."""
    net, indptr, indices, *_ = setup_case(200_000)
    parts = 8
    part_of, sizes = partition_forest(indptr, indices, parts)
    has = net.down_index >= 0
    cut = has & (part_of != part_of[np.maximum(net.down_index, 0)])
    assert cut.sum() > 0 and np.all(part_of[net.down_index[cut]] == parts - 1)
    assert sizes.max() <= 1.02 * 200_000 / parts and sizes.min() >= 0.9 * 200_000 / parts
    n = 4000
    chain_ptr = np.concatenate([np.arange(n), [n - 1]]).astype(np.int32)
    chain_idx = np.arange(1, n, dtype=np.int32)
    part_of, sizes = partition_forest(chain_ptr, chain_idx, 4)
    assert sizes.sum() == n and sizes.max() <= 1.35 * n / 4
    assert np.all(np.diff(part_of) >= 0)


def single_domain(n, T):
    from oracle import oracle
    net, indptr, indices, c1, c2, c3, q0 = setup_case(n)
    ql = synth.synth_qlateral(n, 0, T)
    q, d = q0.copy(), np.zeros((T, n))
    oracle.rapid_route(indptr, indices, -c1[indices], c2, c3, (c1 + c2) / 900.0, q, ql, d, 1)
    return q, d


def test_two_ranks_gloo_match_single_domain(tmp_path):
    import torch.multiprocessing as mp
    n, T, world = 700, 37, 2
    mp.spawn(gloo_worker, args=(world, free_port(), n, T, 8, str(tmp_path)), nprocs=world, join=True)
    q_ref, d_ref = single_domain(n, T)
    q, d = np.zeros(n), np.zeros((T, n))
    for r in range(world):
        z = np.load(tmp_path / f'rank{r}.npz')
        q[z['real']], d[:, z['real']] = z['state'], z['discharge']
    assert_close(q, q_ref, 'state')
    assert_close(d, d_ref, 'discharge')


def test_eight_ranks_gloo_fan_in_matches_single_domain(tmp_path):
    """BASELINE config 5's shape in miniature: eight ranks, seven of them feeding the rank that holds the main stems
    (trunk partition), receives posted a chunk ahead as one batch per chunk -- against the oracle on the undivided network."""
    import torch.multiprocessing as mp
    n, T, world = 6000, 45, 8
    net, indptr, indices, *_ = setup_case(n)
    part_of, _ = partition_forest(indptr, indices, world)
    has = net.down_index >= 0
    cut = has & (part_of != part_of[np.maximum(net.down_index, 0)])
    assert np.all(part_of[net.down_index[cut]] == world - 1) and np.unique(part_of[cut]).size == world - 1      # 7 -> 1
    mp.spawn(gloo_worker, args=(world, free_port(), n, T, 8, str(tmp_path)), nprocs=world, join=True)
    q_ref, d_ref = single_domain(n, T)
    q, d = np.zeros(n), np.zeros((T, n))
    for r in range(world):
        z = np.load(tmp_path / f'rank{r}.npz')
        q[z['real']], d[:, z['real']] = z['state'], z['discharge']
    assert_close(q, q_ref, 'state')
    assert_close(d, d_ref, 'discharge')


def test_in_process_runner_with_oracle_engine_three_parts():
    n, T, parts = 900, 25, 3
    net, indptr, indices, c1, c2, c3, q0 = setup_case(n)
    part_of, _ = partition_forest(indptr, indices, parts)
    specs = [split_network(net.down_index, part_of, p, parts) for p in range(parts)]
    ql = synth.synth_qlateral(n, 0, T)
    engines = [OraclePartEngine(s, c1, c2, c3, (c1 + c2) / 900.0, q0, ql[:, s.real_global], T, 1) for s in specs]
    run_in_process(engines, specs, T, 1, 4)
    q_ref, d_ref = single_domain(n, T)
    q, d = np.zeros(n), np.zeros((T, n))
    for s, e in zip(specs, engines):
        q[s.real_global], d[:, s.real_global] = e.final_state(), e.discharge
    assert_close(q, q_ref, 'state')
    assert_close(d, d_ref, 'discharge')


@pytest.mark.gpu
@pytest.mark.parametrize('wave', ['0', '1', 'rec'])
@pytest.mark.parametrize('n,T,nsub,parts,chunk', [(5000, 40, 1, 3, 8), (5000, 21, 3, 4, 5), (200000, 64, 1, 8, 16),
                                                  (200000, 300, 1, 4, 32),
                                                  (20000, 4000, 1, 3, 64)])     # long enough for the record ring to go round four times
def test_hip_parts_on_one_gpu_match_single_plan(monkeypatch, wave, n, T, nsub, parts, chunk):
    """Ghost/export reaches + rr_stream_* + the driver, on the real engine; reference = one plan over the whole
    network (itself checked against the oracle in test_gpu_kernels.py)."""
    from river_route_amd.engine import Plan
    from river_route_amd.multi_gpu import HipPartEngine
    # boundary reaches in the streaming kernel (k_tick) and in the time-tiled kernel (k_tile) with 16 ticks per task and with
    # the length the engine picks
    monkeypatch.setenv('RR_WAVE', '0' if wave == '0' else '1')
    if wave == '1':
        monkeypatch.setenv('RR_WAVE_K', '16')
    else:
        monkeypatch.delenv('RR_WAVE_K', raising=False)
    net, indptr, indices, c1, c2, c3, q0 = setup_case(n)
    dt = 900.0
    c4_dt = (c1 + c2) / (dt * nsub)
    ql = synth.synth_qlateral(n, 0, T, dt=dt * nsub)
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(-c1[indices], c2, c3, c4_dt)
        q_ref, d_ref = q0.copy(), np.zeros((T, n))
        plan.rapid_route(q_ref, ql, d_ref, nsub)
    part_of, _ = partition_forest(indptr, indices, parts)
    specs = [split_network(net.down_index, part_of, p, parts) for p in range(parts)]
    engines = [HipPartEngine(s, c1, c2, c3, c4_dt, q0, ql[:, s.real_global], T, nsub, 0, out_rows=T) for s in specs]
    run_in_process(engines, specs, T, nsub, chunk)
    q, d = np.zeros(n), np.zeros((T, n))
    for s, e in zip(specs, engines):
        q[s.real_global] = e.final_state()
        d[:, s.real_global] = e.discharge.cpu().numpy()[:, s.n_ghost:]
    assert_close(q, q_ref, 'state')
    assert_close(d, d_ref, 'discharge')
    # second pass on the same engines (bench.py reuses them): identical
    run_in_process(engines, specs, T, nsub, chunk)
    for s, e in zip(specs, engines):
        np.testing.assert_array_equal(e.final_state(), q[s.real_global])


def _ranks_vs_oracle(tmp_path, world, backend, n=120_000, T=200, chunk=32):
    import torch.multiprocessing as mp
    mp.spawn(hip_worker, args=(world, free_port(), n, T, chunk, str(tmp_path), backend), nprocs=world, join=True)
    q_ref, d_ref = single_domain(n, T)
    q, d = np.zeros(n), np.zeros((T, n))
    for r in range(world):
        z = np.load(tmp_path / f'rank{r}.npz')
        q[z['real']], d[:, z['real']] = z['state'], z['discharge']
    assert_close(q, q_ref, 'state')
    assert_close(d, d_ref, 'discharge')


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_over_gloo_vs_oracle(tmp_path):
    """The distributed driver with the real engine: two processes, one card, boundary series over gloo (what a 1-GPU
    box can run of `bench.py --gpus 2`), against the oracle on the undivided network."""
    _ranks_vs_oracle(tmp_path, 2, 'gloo')


@pytest.mark.gpu
def test_five_ranks_share_one_gpu_fan_in_vs_oracle(tmp_path):
    """The shape of `bench.py --gpus 8` as far as one card allows (the GPU box admits six processes on a card, this one
    included): five ranks with the real engine, four leaf parts feeding the trunk part of a 400k-reach network, batched
    receives posted ahead, two passes -- against the oracle on the undivided network."""
    _ranks_vs_oracle(tmp_path, 5, 'gloo', n=400_000, T=160, chunk=32)


@pytest.mark.gpu
def test_two_ranks_rccl_vs_oracle(tmp_path):
    """The same over RCCL (backend 'nccl'), one GPU per rank: needs two GPUs, skipped on a 1-GPU box."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs (RCCL refuses two ranks on one device)')
    _ranks_vs_oracle(tmp_path, 2, 'nccl')


@pytest.mark.gpu
def test_rccl_call_sequence_on_one_rank():
    """Every RCCL call of bench.py's N > 1 leg on the one GPU this box has (a world of one rank, in a child process):
    init with device and timeout, barrier, all_reduce, all_gather, batched irecv / isend of float64 rows, Work.is_completed."""
    import subprocess
    import sys
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port()), RANK='0', WORLD_SIZE='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), 'rccl_one_rank.py')], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'rccl ok nccl' in r.stdout


@pytest.mark.gpu
def test_bench_launches_its_own_ranks_gloo_rehearsal():
    """`python bench.py --gpus 4` with no launcher around it (how the driver starts the bench): the parent starts four ranks before
    it touches the GPU, they share the card over gloo, every rank's part reproduces the oracle on its first 96 rows (the line's
    parity gate) and rank 0's line comes back through the parent."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(RR_DIST_BACKEND='gloo', RR_EXCHANGE_TIMEOUT='240', RR_BENCH_TIMEOUT='900')
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py')
    r = subprocess.run([sys.executable, bench, '--gpus', '4', '--reaches', '100000', '--runoff-steps', '2000', '--steps', '1', '--warmup', '1'],
                       env=env, capture_output=True, text=True, timeout=1000)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 4 and line['config']['reaches'] == 400_000 and line['value'] > 0 and line['scaling'] == 'weak'
    assert sum(line['config']['part_reaches']) == 400_000 and len(line['config']['part_reaches']) == 4
    assert 'oracle' in line['cpu_baseline']['parity_gate']


def test_bench_launcher_reports_a_failed_rank():
    """The launcher returns non-zero and says which rank failed (here: every rank, there is no GPU / bad argument) instead of hanging."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(RR_DIST_BACKEND='gloo', RR_BENCH_TIMEOUT='300')
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py')
    r = subprocess.run([sys.executable, bench, '--gpus', '2', '--reaches', '-5'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and 'exited with code' in r.stderr


def test_exchange_deadline_names_the_missing_message(tmp_path):
    """A receive nobody answers ends the rank with status 1 and a line naming peer and rows, not with a hang."""
    import subprocess
    import sys
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r}); sys.path.insert(0, {os.path.dirname(os.path.abspath(__file__))!r})\n"
        "from river_route_amd.multi_gpu import run_distributed, split_network\n"
        "from river_route_amd.engine import partition_forest\n"
        "from multi_helpers import OraclePartEngine, setup_case\n"
        "from river_route_amd import synth\n"
        "dist.init_process_group('gloo')\n"
        "rank = dist.get_rank()\n"
        "net, indptr, indices, c1, c2, c3, q0 = setup_case(700)\n"
        "part_of, _ = partition_forest(indptr, indices, 2)\n"
        "spec = split_network(net.down_index, part_of, rank, 2)\n"
        "ql = synth.synth_qlateral(700, 0, 16)\n"
        "eng = OraclePartEngine(spec, c1, c2, c3, (c1 + c2) / 900.0, q0, ql[:, spec.real_global], 16, 1)\n"
        "if rank == 0:\n"
        "    import time; time.sleep(20); os._exit(0)      # the upstream part never sends\n"
        "run_distributed(eng, spec, 16, 1, 8, dist)\n")
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RR_EXCHANGE_TIMEOUT='3')
        procs.append(subprocess.Popen([sys.executable, '-c', code], env=env, stderr=subprocess.PIPE, text=True))
    err = procs[1].communicate(timeout=120)[1]
    procs[0].wait(timeout=120)
    assert procs[1].returncode == 1, err
    assert 'receive of boundary sub-steps [0, 8) from part 0 not complete' in err


def test_unit_split_gives_inner_ghosts_a_dummy_headwater():
    """UnitMuskingum on a cut network: a ghost falls on the same side of the headwater / inner distinction
    (river_route/routers/_numba_kernels.py:150-156) as the reach it mirrors."""
    from river_route_amd.multi_gpu import split_network
    n, parts = 20_000, 5
    net = synth.synth_network(n, seed=4)
    has = net.down_index >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
    indices = net.down_index[has].astype(np.int32)
    part_of, _ = partition_forest(indptr, indices, parts)
    inner = np.bincount(net.down_index[has], minlength=n) > 0
    # move one headwater away from the reach it flows into, to an upstream part: a cut directly below a headwater
    hw = np.flatnonzero(~inner & has & (part_of == parts - 1))
    part_of = part_of.copy()
    part_of[hw[0]] = 0
    seen_hw_ghost = False
    for p in range(parts):
        spec = split_network(net.down_index, part_of, p, parts, inner_global=inner)
        nd, ng = spec.n_dummy, spec.n_ghost
        assert nd == int(inner[spec.ghost_global].sum())
        local_inner = np.bincount(spec.down_local[spec.down_local >= 0], minlength=spec.n_local) > 0
        np.testing.assert_array_equal(local_inner[nd:nd + ng], inner[spec.ghost_global])       # ghosts: as their reaches
        np.testing.assert_array_equal(local_inner[nd + ng:], inner[spec.real_global])           # own reaches: unchanged
        assert not local_inner[:nd].any()                                                       # dummies are headwaters
        np.testing.assert_array_equal(spec.down_local[:nd], nd + spec.dummy_ghost)
        assert (spec.down_local[nd:nd + ng] >= nd + ng).all()                                   # a ghost flows into an own reach
        seen_hw_ghost |= bool((~inner[spec.ghost_global]).any())
        plain = split_network(net.down_index, part_of, p, parts)
        assert plain.n_dummy == 0 and plain.n_local == spec.n_local - nd
    assert seen_hw_ghost

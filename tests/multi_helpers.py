"""Test-only helpers for the partitioned (multi-GPU) path: an oracle-backed stand-in for HipPartEngine so the
partitioning + exchange logic runs under gloo on CPU, and the worker the spawned ranks execute."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def setup_case(n, seed=4):
    from river_route_amd import synth
    from oracle import oracle
    net = synth.synth_network(n, seed=seed)
    has = net.down_index >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
    indices = net.down_index[has].astype(np.int32)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0)
    q0 = 4.0 * synth.u01(8, np.arange(n))
    return net, indptr, indices, c1, c2, c3, q0


class OraclePartEngine:
    """Same interface as river_route_amd.multi_gpu.HipPartEngine, computed by oracle/ (nsub = 1 only: the ghost
    inflow c2*Q[t-1] + c1*Q[t] is folded into the lateral volume of the receiving reach)."""

    def __init__(self, spec, c1, c2, c3, c4_dt, q0_global, lateral_rows, T, nsub):
        import torch
        assert nsub == 1
        self.spec, self.T = spec, T
        real = spec.real_global
        self.c1, self.c2, self.c3, self.c4 = c1[real], c2[real], c3[real], c4_dt[real]
        ng = spec.n_ghost
        down = spec.down_local[ng:] - ng          # real -> real local downstream (ghosts are never downstream)
        down = np.where(spec.down_local[ng:] >= 0, down, -1)
        has = down >= 0
        self.indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
        self.indices = down[has].astype(np.int32)
        self.lhs = -self.c1[self.indices]
        self.ghost_target = spec.down_local[:ng] - ng          # real reach each ghost flows into
        self.lateral = lateral_rows
        self.q0_real = q0_global[real].copy()
        self.q0_ghost = q0_global[spec.ghost_global].copy()
        self.export_local = np.searchsorted(real, spec.export_global)
        self.ghost_series = torch.zeros((T, max(ng, 1)), dtype=torch.float64)
        self.export_series = torch.zeros((T, max(spec.export_global.size, 1)), dtype=torch.float64)
        self.discharge = np.zeros((T, real.size))

    def begin(self):
        self.q = self.q0_real.copy()
        self.done = 0

    def advance(self, rows_ready, ghost_ready):
        from oracle import oracle
        ng = self.spec.n_ghost
        stop = min(rows_ready, ghost_ready if ng else self.T, self.T)
        G = self.ghost_series.numpy()
        for t in range(self.done, stop):
            ql = self.lateral[t % self.lateral.shape[0]].copy()
            for g in range(ng):
                d = self.ghost_target[g]
                old = self.q0_ghost[g] if t == 0 else G[t - 1, g]
                ql[d] += (self.c2[d] * old + self.c1[d] * G[t, g]) / self.c4[d]
            row = np.zeros((1, self.q.size))
            oracle.rapid_route(self.indptr, self.indices, self.lhs, self.c2, self.c3, self.c4, self.q, ql[None, :], row, 1)
            self.discharge[t] = row[0]
            if self.export_local.size:
                self.export_series[t, :self.export_local.size] = __import__('torch').from_numpy(self.q[self.export_local])
        self.done = max(self.done, stop)
        return self.done

    def end(self):
        assert self.done == self.T

    def final_state(self):
        return self.q


def gloo_worker(rank, world, port, n, T, chunk_rows, out_dir):
    import torch.distributed as dist
    from river_route_amd import synth
    from river_route_amd.engine import partition_forest
    from river_route_amd.multi_gpu import run_distributed, split_network
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    net, indptr, indices, c1, c2, c3, q0 = setup_case(n)
    part_of, _ = partition_forest(indptr, indices, world)
    spec = split_network(net.down_index, part_of, rank, world)
    ql = synth.synth_qlateral(n, 0, T)
    eng = OraclePartEngine(spec, c1, c2, c3, (c1 + c2) / 900.0, q0, ql[:, spec.real_global], T, 1)
    run_distributed(eng, spec, T, 1, chunk_rows, dist)
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), real=spec.real_global, discharge=eng.discharge,
             state=eng.final_state())
    dist.barrier()
    dist.destroy_process_group()


def hip_worker(rank, world, port, n, T, chunk_rows, out_dir, backend):
    """One rank of the real partitioned run: HipPartEngine on GPU `rank % device_count`, boundary series over `backend`
    ('nccl' = RCCL, one GPU per rank; 'gloo' lets the ranks of a rehearsal share one card)."""
    import torch
    import torch.distributed as dist
    from river_route_amd import synth
    from river_route_amd.engine import partition_forest
    from river_route_amd.multi_gpu import HipPartEngine, run_distributed, split_network
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    device = rank % torch.cuda.device_count()
    torch.cuda.set_device(device)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', device))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    net, indptr, indices, c1, c2, c3, q0 = setup_case(n)
    part_of, _ = partition_forest(indptr, indices, world)
    spec = split_network(net.down_index, part_of, rank, world)
    ql = synth.synth_qlateral(n, 0, T)
    eng = HipPartEngine(spec, c1, c2, c3, (c1 + c2) / 900.0, q0, ql[:, spec.real_global], T, 1, device, out_rows=T)
    dist.barrier()              # as bench.py: the group's communicator exists before the first batched send / receive
    for _ in range(2):          # bench.py reuses the engines pass after pass
        run_distributed(eng, spec, T, 1, chunk_rows, dist)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'), real=spec.real_global, state=eng.final_state(),
             discharge=eng.discharge.cpu().numpy()[:, spec.n_ghost:])
    dist.barrier()
    dist.destroy_process_group()

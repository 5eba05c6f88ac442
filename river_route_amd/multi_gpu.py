"""
Graph-partitioned routing across the GPUs of one node (BASELINE config 5).

The river network is cut into balanced parts, one per GPU (`rr_partition_forest`: main stems and their farthest
tributaries in the last part, the other subtrees spread over the rest; parts numbered upstream-first, the part
graph is acyclic because discharge only flows downstream).  For every cut
edge (u -> d) the part that owns `d` carries `u` as a GHOST reach whose discharge series is prescribed, and the
part that owns `u` records `u`'s discharge after every routing sub-step in its export series
(`rr_plan_set_boundary`).  The only data-path communication is that series, sent downstream in batches of
`chunk_rows` runoff steps with point-to-point `torch.distributed` send/recv (RCCL over xGMI on the GPU box,
gloo in the CPU tests): a few KB per batch, so the exchange is latency-bound and batching hides it.  Each
GPU keeps its lag pipeline open across batches (`rr_stream_begin/advance/end`), so a downstream part simply
runs a fixed number of ticks behind its upstream parts and all GPUs compute concurrently.

UnitMuskingum crosses the cut the same way (`HipUnitPartEngine`): what travels is the discharge a reach publishes
(q_full; the convolved lateral on a headwater).  Its kernel treats headwater tributaries differently from the others
(river_route/routers/_numba_kernels.py:150-156), so a ghost that mirrors a reach WITH upstream reaches gets a dummy
headwater above it in the local network (zero lateral, value never used): the ghost then counts as an inner tributary,
exactly as the reach it mirrors does in the uncut network, while a ghost that mirrors a headwater stays a headwater.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

# RCCL shares device memory between the ranks of a node through dmabuf IPC on these hosts; the legacy mode fails with
# `hipIpcGetMemHandle: invalid argument`.  The variable is read when the HIP runtime starts: bench.py's launcher puts it into the ranks'
# environment; a program that drives run_distributed itself sets it before it touches the GPU.  Setting it here only helps when this
# module is imported first, so say so when it is too late.
if os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY') is None:
    os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    import sys as _sys
    _t = _sys.modules.get('torch')
    if _t is not None and getattr(_t, 'cuda', None) is not None and _t.cuda.is_initialized():
        import warnings
        warnings.warn('river_route_amd.multi_gpu: HSA_ENABLE_IPC_MODE_LEGACY was not set before the HIP runtime started; RCCL over several '
                      'GPUs needs HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment of the process (hipIpcGetMemHandle fails otherwise)')

__all__ = ['PartSpec', 'split_network', 'HipPartEngine', 'HipUnitPartEngine', 'part_driver', 'run_distributed', 'run_in_process',
           'run_sequential', 'bench_main']


@dataclass
class PartSpec:
    part: int
    n_parts: int
    real_global: np.ndarray          # global params indices of this part's reaches, ascending
    ghost_global: np.ndarray         # global index of the upstream reach behind each ghost
    ghost_owner: np.ndarray          # part that owns it
    export_global: np.ndarray        # global indices of this part's export reaches
    export_consumer: np.ndarray      # part that consumes each export
    indptr: np.ndarray               # local CSC (ghosts first, then real reaches), int32
    indices: np.ndarray
    down_local: np.ndarray           # local downstream index, -1 at local outlets
    upstream_parts: list = field(default_factory=list)    # [(part, ghost column slice)]
    downstream_parts: list = field(default_factory=list)  # [(part, export column slice)]
    dummy_ghost: np.ndarray = field(default_factory=lambda: np.zeros(0, dtype=np.int64))   # UnitMuskingum: ghost (by number) below each dummy headwater

    @property
    def n_ghost(self) -> int:
        return int(self.ghost_global.size)

    @property
    def n_dummy(self) -> int:
        return int(self.dummy_ghost.size)

    @property
    def n_lead(self) -> int:
        """Local columns before the part's own reaches: dummies, then ghosts."""
        return self.n_dummy + self.n_ghost

    @property
    def n_local(self) -> int:
        return int(self.n_lead + self.real_global.size)


def split_network(down_index: np.ndarray, part_of: np.ndarray, part: int, n_parts: int, inner_global=None) -> PartSpec:
    """Local network of one part.  Local index order: ghosts (grouped by owning part, ascending global index)
    first, then the part's reaches in ascending global index -- still upstream before downstream.  With `inner_global`
    (UnitMuskingum: True where a reach of the uncut network has upstream reaches) every ghost that mirrors such a reach
    gets a dummy headwater above it; the dummies come before the ghosts."""
    down_index = np.asarray(down_index, dtype=np.int64)
    part_of = np.asarray(part_of)
    real = np.flatnonzero(part_of == part)
    has_down = down_index >= 0
    dpart = np.where(has_down, part_of[np.maximum(down_index, 0)], -1)
    cut_in = np.flatnonzero(has_down & (dpart == part) & (part_of != part))      # upstream ends, owned elsewhere
    cut_in = cut_in[np.lexsort((cut_in, part_of[cut_in]))]
    cut_out = np.flatnonzero(has_down & (part_of == part) & (dpart != part))
    cut_out = cut_out[np.lexsort((cut_out, dpart[cut_out]))]
    ng = cut_in.size
    dummy_ghost = np.flatnonzero(np.asarray(inner_global, dtype=bool)[cut_in]) if inner_global is not None else np.zeros(0, dtype=np.int64)
    nd = dummy_ghost.size
    local_of = np.full(down_index.size, -1, dtype=np.int64)
    local_of[cut_in] = nd + np.arange(ng)
    local_of[real] = nd + ng + np.arange(real.size)
    members = np.concatenate([cut_in, real])
    down_local = np.concatenate([nd + dummy_ghost, np.where(has_down[members], local_of[np.maximum(down_index[members], 0)], -1)])
    down_local[nd + ng:][np.isin(real, cut_out)] = -1          # exports are outlets of the local network
    has = down_local >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
    indices = down_local[has].astype(np.int32)
    spec = PartSpec(part, n_parts, real, cut_in, part_of[cut_in].astype(np.int64), cut_out,
                    dpart[cut_out].astype(np.int64), indptr, indices, down_local, dummy_ghost=dummy_ghost.astype(np.int64))
    for owners, out in ((spec.ghost_owner, spec.upstream_parts), (spec.export_consumer, spec.downstream_parts)):
        for p in np.unique(owners):
            cols = np.flatnonzero(owners == p)
            out.append((int(p), slice(int(cols[0]), int(cols[-1]) + 1)))
    return spec


class HipPartEngine:
    """One part on one GPU: plan + coefficients + device-resident forcing, state and boundary series (torch
    tensors for memory and streams only; all compute is librr_hip.so)."""

    def __init__(self, spec: PartSpec, c1, c2, c3, c4_dt, q0_global, lateral_rows, T, nsub, device, out_rows=None,
                 sample_every=0):
        import torch
        from .engine import Plan
        self.torch = torch
        self.spec, self.T, self.nsub = spec, int(T), int(nsub)
        self.dev = torch.device('cuda', device)
        ng, members = spec.n_ghost, np.concatenate([spec.ghost_global, spec.real_global])
        n_loc = members.size

        def local(v, ghost_value=0.0):
            a = np.asarray(v, dtype=np.float64)[members].copy()
            a[:ng] = ghost_value
            return a

        self.plan = Plan(spec.indptr, spec.indices, device=device)
        c1_loc = np.asarray(c1, dtype=np.float64)[members]
        # off-diagonal of the edge leaving local reach j is -c1 of its downstream reach; for a ghost that is the
        # real downstream reach across the cut
        has = spec.down_local >= 0
        lhs = -c1_loc[spec.down_local[has]]
        self.plan.set_coeffs(lhs, local(c2), local(c3), None if c4_dt is None else local(c4_dt))
        export_local = ng + np.searchsorted(spec.real_global, spec.export_global)
        self.plan.set_boundary(np.arange(ng), export_local)
        self.plan.set_options(sample_every=sample_every)
        # lateral rows: (rows, n_loc) in local order; ghost columns are never read
        self.lat_rows = int(lateral_rows.shape[0])
        self.lateral = torch.zeros((self.lat_rows, n_loc), dtype=torch.float64, device=self.dev)
        self.lateral[:, ng:] = lateral_rows.to(self.dev) if torch.is_tensor(lateral_rows) else torch.from_numpy(np.ascontiguousarray(lateral_rows, dtype=np.float64)).to(self.dev)
        self.out_rows = int(out_rows or self.lat_rows)
        self.discharge = torch.zeros((self.out_rows, n_loc), dtype=torch.float64, device=self.dev)
        q0 = np.asarray(q0_global, dtype=np.float64)[members]      # ghosts start from their reach's own state
        self.q0 = torch.from_numpy(q0).to(self.dev)
        self.q_t = torch.empty_like(self.q0)
        S = self.T * self.nsub
        self.ghost_series = torch.zeros((S, max(ng, 1)), dtype=torch.float64, device=self.dev)
        self.export_series = torch.zeros((S, max(spec.export_global.size, 1)), dtype=torch.float64, device=self.dev)

    def begin(self) -> None:
        self.q_t.copy_(self.q0)
        stream = self.torch.cuda.current_stream(self.dev).cuda_stream
        self.plan.stream_begin(self.q_t, self.lateral, self.lat_rows, self.discharge, self.out_rows, self.T,
                               self.nsub, self.ghost_series, self.export_series, stream)

    def reshape_call(self, T: int, discharge=None, out_rows=None) -> None:
        """The next calls route T runoff steps (at most the T the engine was built for: the boundary series are that long)
        into `discharge` (a (out_rows, n_local) device tensor written cyclically) -- bench.py's parity gate routes the first rows
        into a plain array with the engine it then times."""
        if T * self.nsub > self.ghost_series.shape[0]:
            raise ValueError('reshape_call: more steps than the boundary series hold')
        self.T = int(T)
        if discharge is not None:
            self.discharge, self.out_rows = discharge, int(out_rows or discharge.shape[0])

    def advance(self, rows_ready: int, ghost_ready: int) -> int:
        return self.plan.stream_advance(rows_ready, ghost_ready)

    def end(self) -> None:
        self.plan.stream_end(self.q_t)

    def final_state(self) -> np.ndarray:
        return self.q_t.cpu().numpy()[self.spec.n_ghost:]

    def close(self) -> None:
        """Hands the plan's device memory (record ring included) and the part's rows back; the export series stays with whoever
        holds it (run_sequential: the parts downstream)."""
        self.plan.close()
        self.lateral = self.discharge = self.ghost_series = self.q0 = self.q_t = None
        self.torch.cuda.empty_cache()


class HipUnitPartEngine:
    """One part of a UnitMuskingum run on one GPU (river_route/routers/UnitMuskingum.py:72-98 over a cut network): the
    part convolves its own columns (the convolution is column-wise, rr_uh_convolve_dev) and routes them with
    rr_stream_begin_unit / rr_stream_advance / rr_stream_end_unit; ghosts carry the discharge their reaches publish.

    `inner_global`: True where a reach of the uncut network has upstream reaches; q_ch / q_full: state over those reaches
    (ascending global index), as the reference's router keeps it."""

    def __init__(self, spec: PartSpec, c1, c2, c3, inner_global, q_ch, q_full, uh_kernel, uh_state, depth_rows, T, nsub, device):
        import torch
        from .engine import Plan, uh_convolve_dev
        self.torch = torch
        self.spec, self.T, self.nsub = spec, int(T), int(nsub)
        self.dev = torch.device('cuda', device)
        nd, ng, lead = spec.n_dummy, spec.n_ghost, spec.n_lead
        inner_global = np.asarray(inner_global, dtype=bool)
        members = np.concatenate([spec.ghost_global[spec.dummy_ghost], spec.ghost_global, spec.real_global])   # a dummy borrows its ghost's reach
        n_loc = members.size
        self.plan = Plan(spec.indptr, spec.indices, device=device)
        has = spec.down_local >= 0
        c1_loc = np.asarray(c1, dtype=np.float64)[members]
        self.plan.set_coeffs(-c1_loc[spec.down_local[has]], np.asarray(c2, dtype=np.float64)[members],
                             np.asarray(c3, dtype=np.float64)[members], None)
        self.plan.set_boundary(nd + np.arange(ng), lead + np.searchsorted(spec.real_global, spec.export_global))
        # state over the local reaches that have upstream reaches: ghosts below a dummy (q_full = their reach's), then real ones
        self.inner_local = np.flatnonzero(np.bincount(spec.down_local[has], minlength=n_loc) > 0)
        assert self.inner_local.size == self.plan.n_inner
        rank = np.cumsum(inner_global) - 1
        self.inner_rank = rank[members[self.inner_local]]
        self.q_ch0 = torch.from_numpy(np.asarray(q_ch, dtype=np.float64)[self.inner_rank]).to(self.dev)
        self.q_full0 = torch.from_numpy(np.asarray(q_full, dtype=np.float64)[self.inner_rank]).to(self.dev)
        self.q_ch, self.q_full = torch.empty_like(self.q_ch0), torch.empty_like(self.q_full0)
        # convolution of this part's columns on the device; dummy and ghost columns of the lateral rows are zero and unread
        rows, n_real = depth_rows.shape
        n_ks = uh_kernel.shape[0]
        d_depth = torch.from_numpy(np.ascontiguousarray(depth_rows, dtype=np.float64)).to(self.dev)
        d_kernel = torch.from_numpy(np.ascontiguousarray(uh_kernel, dtype=np.float64)).to(self.dev)
        self.uh_state = torch.from_numpy(np.ascontiguousarray(uh_state, dtype=np.float64)).to(self.dev)
        conv = torch.empty_like(d_depth)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        uh_convolve_dev(d_kernel, self.uh_state, d_depth, conv, rows, n_ks, n_real, device=device, stream=stream)
        self.lateral = torch.zeros((rows, n_loc), dtype=torch.float64, device=self.dev)
        self.lateral[:, lead:] = conv
        self.lat_rows = rows
        self.discharge = torch.zeros((rows, n_loc), dtype=torch.float64, device=self.dev)
        S = self.T * self.nsub
        self.ghost_series = torch.zeros((S, max(ng, 1)), dtype=torch.float64, device=self.dev)
        self.export_series = torch.zeros((S, max(spec.export_global.size, 1)), dtype=torch.float64, device=self.dev)

    def begin(self) -> None:
        self.q_ch.copy_(self.q_ch0)
        self.q_full.copy_(self.q_full0)
        stream = self.torch.cuda.current_stream(self.dev).cuda_stream
        self.plan.stream_begin_unit(self.q_ch, self.q_full, self.lateral, self.lat_rows, self.discharge, self.lat_rows, self.T,
                                    self.nsub, self.ghost_series, self.export_series, stream)

    def advance(self, rows_ready: int, ghost_ready: int) -> int:
        return self.plan.stream_advance(rows_ready, ghost_ready)

    def end(self) -> None:
        self.plan.stream_end_unit(self.q_ch, self.q_full)

    def final_state(self):
        """(global inner rank, q_ch, q_full) of this part's own reaches."""
        own = self.inner_local >= self.spec.n_lead
        return self.inner_rank[own], self.q_ch.cpu().numpy()[own], self.q_full.cpu().numpy()[own]


def part_driver(engine, spec: PartSpec, T: int, nsub: int, chunk_rows: int):
    """
    Generator running one part through one routing call.  It yields requests the runner must serve:
      ('post', k, [(src_part, ghost_view, r0, r1), ...])   rows [r0, r1) of each upstream part's export block will be needed
                                                             for chunk k: receives may be posted now (one chunk ahead of use)
      ('recv', src_part, ghost_view, r0, r1)               block until those rows arrived and sit in ghost_view
      ('send', dst_part, tensor, r0, r1)                   ship rows [r0, r1) of the export block for dst
    """
    S = T * nsub
    kc = chunk_rows * nsub
    engine.begin()
    sent = 0
    n_chunks = (T + chunk_rows - 1) // chunk_rows

    def flush(export_ready, final):
        nonlocal sent
        while sent < S and (export_ready - sent >= kc or (final and export_ready > sent) or
                            (export_ready >= S and sent < S)):
            r1 = min(sent + kc, export_ready, S)
            for dst, cols in spec.downstream_parts:
                yield ('send', dst, engine.export_series[sent:r1, cols], sent, r1)
            sent = r1

    def wanted(k):
        s0, s1 = k * kc, min(T, (k + 1) * chunk_rows) * nsub
        return [(src, engine.ghost_series[s0:s1, cols], s0, s1) for src, cols in spec.upstream_parts]

    if spec.upstream_parts and n_chunks:
        yield ('post', 0, wanted(0))
    for k in range(n_chunks):
        rows1 = min(T, (k + 1) * chunk_rows)
        s1 = rows1 * nsub
        for req in wanted(k):
            yield ('recv',) + req
        ready = engine.advance(rows1, s1 if spec.n_ghost else S)
        if spec.downstream_parts:
            yield from flush(ready, False)
        # The next chunk's receives are posted after this chunk's sends: on RCCL both queue on the communicator's stream, and a
        # receive posted first would hold this part's exports back until ITS upstream parts have produced the next chunk (one
        # chunk of latency per level of the part graph).  The host never blocks on RCCL work, so they are still posted long before
        # the values are used.
        if spec.upstream_parts and k + 1 < n_chunks:
            yield ('post', k + 1, wanted(k + 1))
    ready = engine.advance(T, S)
    if spec.downstream_parts:
        yield from flush(ready, True)
    engine.end()


def run_distributed(engine, spec: PartSpec, T: int, nsub: int, chunk_rows: int, dist) -> None:
    """Serve one part's driver with torch.distributed point-to-point ops (nccl = RCCL on the GPU box, gloo on CPU).
    Rank == part.  The receives of a chunk are posted as ONE batch (one grouped RCCL call for all upstream parts) when the
    driver announces them -- a chunk before the values are used -- so the part that collects seven boundary series never
    stands in a queue of blocking receives; sends are asynchronous (batches of one, so that both ends of a message use the same
    communicator), their buffers kept alive until the call ends.

    Nothing here waits without a deadline (RR_EXCHANGE_TIMEOUT seconds, default 300): a message that does not arrive -- a
    mismatched pair, a rank that died -- is reported with its peer and rows and the process exits with status 1, so the
    launcher (bench.py / torchrun) takes the other ranks down instead of every rank hanging until its lease ends."""
    import os
    import sys
    import time
    limit = float(os.environ.get('RR_EXCHANGE_TIMEOUT', '300'))
    rank = dist.get_rank()
    pending = []                                # sends: (work, what, posted at, buffer kept alive)
    posted = {}                                 # (src, r0, r1) -> (work, staging buffer, posted at)
    watch = []                                  # RCCL receives the stream waits for: (work, what, posted at)
    via_host = dist.get_backend() != 'nccl'     # gloo moves host memory: stage device tensors through the CPU

    def give_up(what, since):
        print(f'rr: rank {rank}: {what} not complete after {time.monotonic() - since:.0f} s (RR_EXCHANGE_TIMEOUT={limit:g}); '
              f'still open: {len(posted)} receive batch(es), {len(pending)} send(s) -- exiting', file=sys.stderr, flush=True)
        os._exit(1)       # not sys.exit: interpreter teardown would wait for the communicator the stuck operation holds

    def complete(work, what, since, block):
        """True once `work` is done; with `block`, waits until then or until the deadline."""
        if via_host:      # gloo's point-to-point work knows no is_completed(); its wait takes a timeout and raises when that passes
            if not block:
                return False
            import datetime
            try:
                work.wait(datetime.timedelta(seconds=max(0.05, limit - (time.monotonic() - since))))
            except RuntimeError:
                give_up(what, since)
            return True
        while not work.is_completed():      # RCCL: a query of the event behind the operation
            if time.monotonic() - since > limit:
                give_up(what, since)
            if not block:
                return False
            time.sleep(0.0005)
        return True

    def sweep(block=False):
        while watch and complete(*watch[0], block):
            watch.pop(0)
        while pending and complete(*pending[0][:3], block):
            pending.pop(0)

    for req in part_driver(engine, spec, T, nsub, chunk_rows):
        kind = req[0]
        if kind == 'post':
            ops, keys = [], []
            for src, view, r0, r1 in req[2]:
                buf = view.new_empty(view.shape, device='cpu') if via_host else view.new_empty(view.shape)
                ops.append(dist.P2POp(dist.irecv, buf, src))
                keys.append(((src, r0, r1), buf))
            works = dist.batch_isend_irecv(ops)
            now = time.monotonic()
            for i, (key, buf) in enumerate(keys):
                posted[key] = (works[i] if len(works) == len(keys) else works[-1], buf, now)      # RCCL returns one work for the group
        elif kind == 'recv':
            _, peer, view, r0, r1 = req
            work, buf, since = posted.pop((peer, r0, r1))
            what = f'receive of boundary sub-steps [{r0}, {r1}) from part {peer}'
            if via_host:
                complete(work, what, since, True)      # the copy below reads host memory
            else:
                work.wait()                            # RCCL: the current stream waits, the host does not ...
                watch.append((work, what, since))      # ... so the host looks again later
            view.copy_(buf)
        else:
            _, peer, view, r0, r1 = req
            buf = view.cpu().contiguous() if via_host else view.contiguous()
            # a batch of one, not dist.isend: ProcessGroupNCCL sends a batched operation over the group's communicator and a single
            # one over a two-rank communicator of its own, and a receive posted in a batch (above) only ever meets the former
            work = dist.batch_isend_irecv([dist.P2POp(dist.isend, buf, peer)])[-1]
            pending.append((work, f'send of boundary sub-steps [{r0}, {r1}) to part {peer}', time.monotonic(), buf))
        sweep()
    assert not posted
    sweep(block=True)


def run_in_process(engines, specs, T: int, nsub: int, chunk_rows: int) -> None:
    """All parts in one process (one GPU): a cooperative scheduler with mailboxes instead of a network.  Used by
    the single-GPU tests to exercise exactly the driver and engine paths the distributed run uses."""
    drivers = [part_driver(e, s, T, nsub, chunk_rows) for e, s in zip(engines, specs)]
    mail: dict = {}
    waiting = [None] * len(drivers)
    alive = set(range(len(drivers)))
    while alive:
        progressed = False
        for p in sorted(alive):
            while True:
                req = waiting[p]
                if req is None:
                    try:
                        req = next(drivers[p])
                    except StopIteration:
                        alive.discard(p)
                        progressed = True
                        break
                if req[0] == 'post':            # nothing to post without a network: the mailbox is the network
                    waiting[p] = None
                    progressed = True
                    continue
                kind, peer, view, r0, r1 = req
                if kind == 'send':
                    mail.setdefault((p, peer), []).append((r0, r1, view.clone()))
                    waiting[p] = None
                    progressed = True
                    continue
                box = mail.get((peer, p), [])
                if box and box[0][0] == r0 and box[0][1] == r1:
                    view.copy_(box.pop(0)[2])
                    waiting[p] = None
                    progressed = True
                    continue
                waiting[p] = req
                break
        if not progressed:
            raise RuntimeError('run_in_process: parts are deadlocked (no message can be delivered)')


def run_sequential(specs, make_engine, T: int, nsub: int, visit=None, route=None) -> None:
    """The parts of a cut network routed ONE AFTER ANOTHER on one GPU.  Discharge only flows downstream
    (docs/references/math.md:57-59, docs/references/parallelism.md:67-75), and the parts are numbered upstream-first, so part p
    can be routed through its whole call once the parts before it have been: their export series (kept on the device, a few MB)
    are its ghost series.  `specs`: every part's PartSpec in part order; `make_engine(spec)` builds the part's engine (its plan
    and record ring exist only while the part is routed: a network whose parts each fit the card routes on it whatever its
    size); `visit(spec, engine)` sees the finished engine before it is closed; `route(engine, T, S)` replaces the one
    begin / advance / end (bench.py times several passes per part).  What bench.py --sequential-parts and the BASELINE config 5
    test run: the same engines, kernels and boundary series as the distributed run, without the exchange."""
    S = int(T) * int(nsub)
    exports = {}
    for spec in specs:
        if any(src >= spec.part for src, _ in spec.upstream_parts):
            raise ValueError('run_sequential: parts must be numbered upstream-first')
        eng = make_engine(spec)
        for src, cols in spec.upstream_parts:
            ecols = next(c for dst, c in specs[src].downstream_parts if dst == spec.part)
            eng.ghost_series[:S, cols] = exports[src][:S, ecols]
        if route is not None:
            route(eng, int(T), S)
        else:
            eng.begin()
            eng.advance(T, S)
            eng.end()
        if spec.downstream_parts:
            exports[spec.part] = eng.export_series
        if visit is not None:
            visit(spec, eng)
        if hasattr(eng, 'close'):
            eng.close()


# ------------------------------------------------------------------------------------------------ bench.py --gpus N

def bench_main(args, rank: int, local_rank: int, world: int, gate=None) -> None:
    """N > 1 leg of bench.py: ONE network of args.reaches * N reaches, graph-partitioned over the N GPUs
    (weak scaling: reaches per GPU fixed), boundary discharge exchanged over RCCL.

    `gate(eng, spec, coefficients, run)` -> str: bench.py's parity gate (it owns the oracle: nothing in this package imports
    it).  It routes the first rows through `run` -- the distributed driver that is then timed -- and compares this part's rows
    with the oracle; every rank must pass before anything is timed."""
    import json
    import time
    import torch
    import torch.distributed as dist
    from . import synth
    from .engine import partition_forest

    n, T, nsub, dt = args.reaches * world, args.runoff_steps, args.substeps, 900.0      # 1.25M per GPU by default: 10M on 8 GPUs (BASELINE config 5)
    # Every rank builds the network and the partition itself: both are deterministic functions of (n, seed), 21 s of one host
    # core at 10M reaches (17.6 s the synthetic network, 3.7 s the partition; profiles/r03_multi_setup.txt) against shipping
    # 10M x 4 arrays from rank 0; `setup_s` in the line says what it cost on the box.
    t_setup = time.perf_counter()
    net = synth.synth_network(n, order=args.order)
    has = net.down_index >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
    indices = net.down_index[has].astype(np.int32)
    t_net = time.perf_counter()
    part_of, sizes = partition_forest(indptr, indices, world)
    spec = split_network(net.down_index, part_of, rank, world)
    t_part = time.perf_counter()
    r = dt / net.k
    den = r + 2.0 * (1.0 - net.x)
    c1, c2, c3 = (r - 2.0 * net.x) / den, (r + 2.0 * net.x) / den, (2.0 * (1.0 - net.x) - r) / den
    rows = min(args.forcing_rows or 288, T)
    lateral = synth.synth_qlateral_torch(n, 0, rows, torch.device('cuda', local_rank), columns=spec.real_global, dt=dt * nsub)
    eng = HipPartEngine(spec, c1, c2, c3, (c1 + c2) / (dt * nsub), np.zeros(n), lateral, T, nsub, local_rank,
                        out_rows=min(T, 128), sample_every=args.sample_every)      # sink of one out-pass batch: no row written twice by a launch
    chunk_rows = max(16, args.exchange_rows)
    from .engine import MODE_RAPID
    sched = eng.plan.reserve(MODE_RAPID, T, nsub)      # the record ring, before anything is timed
    torch.cuda.synchronize()
    dist.barrier()      # every rank is through its setup, and the group's communicator exists before the first batched send / receive uses it
    t_ready = time.perf_counter()
    cdev = eng.dev if dist.get_backend() == 'nccl' else torch.device('cpu')

    gate_note = None
    if gate is not None:
        sink, sink_rows = eng.discharge, eng.out_rows
        ok, note = 1.0, ''
        try:
            note = gate(eng, spec, (c1, c2, c3, (c1 + c2) / (dt * nsub)), lambda Tg: run_distributed(eng, spec, Tg, nsub, min(chunk_rows, 32), dist))
        except AssertionError as e:
            ok, note = 0.0, str(e)
        eng.reshape_call(T, sink, sink_rows)
        flag = torch.tensor([ok], dtype=torch.float64, device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if ok == 0.0:
            print(f'bench.py: rank {rank}: partitioned result differs from the oracle: {note}', file=__import__('sys').stderr, flush=True)
        if float(flag.item()) == 0.0:
            raise SystemExit('bench.py: a part of the partitioned run does not reproduce the oracle; refusing to report a number')
        gate_note = note

    def one_pass():
        run_distributed(eng, spec, T, nsub, chunk_rows, dist)

    for _ in range(args.warmup):
        one_pass()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=cdev)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    prof = eng.plan.profile()
    tiles = eng.plan.tile_info()
    info = torch.tensor([float(spec.real_global.size), float(spec.n_ghost), float(eng.plan.depth), sched['ring_bytes'] / 1e9,
                         float(sched['ticks_per_launch']), float(tiles['levels']), t_net - t_setup, t_part - t_net, t_ready - t_part,
                         prof['region_ms']], dtype=torch.float64, device=cdev)
    gathered = [torch.zeros_like(info) for _ in range(world)]
    dist.all_gather(gathered, info)
    if rank == 0:
        from .measure import roofline_from_profile
        roofline = roofline_from_profile(prof, nsub)
        line = {
            'metric': 'reach-steps/sec', 'value': float(n) * T * nsub * args.steps / elapsed,
            'unit': 'reach-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'RapidMuskingum, ONE {n}-reach synthetic random-topology network graph-partitioned '
                                   f'over {world} MI355X ({args.reaches} reaches per GPU), {T} runoff steps @ 900 s, '
                                   f'{nsub} sub-step(s), fp64, boundary discharge exchanged over '
                                   f'{"RCCL (xGMI)" if dist.get_backend() == "nccl" else dist.get_backend()} every {chunk_rows} steps',
                       'reaches': n, 'runoff_steps': T, 'substeps': nsub, 'params_order': args.order,
                       'part_reaches': [int(g[0].item()) for g in gathered],
                       'part_ghosts': [int(g[1].item()) for g in gathered],
                       'part_depth': [int(g[2].item()) for g in gathered],
                       'part_ring_gb': [round(g[3].item(), 1) for g in gathered],
                       'part_ticks_per_launch': [int(g[4].item()) for g in gathered],
                       'part_tile_levels': [int(g[5].item()) for g in gathered],
                       'part_pass_ms': [round(g[9].item(), 1) for g in gathered],      # first to last routing launch of the last pass, per GPU
                       'setup_s': {'network': [round(g[6].item(), 1) for g in gathered], 'partition': [round(g[7].item(), 1) for g in gathered],
                                   'plan_forcing_ring': [round(g[8].item(), 1) for g in gathered]},
                       'exchange_rows': chunk_rows},
            'roofline': roofline, 'cpu_baseline': None if gate_note is None else {'parity_gate': f'every rank: {gate_note}'},
        }
        print(json.dumps(line), flush=True)
    dist.destroy_process_group()

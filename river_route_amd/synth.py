"""
Seeded synthetic river networks, routing parameters and forcing for tests and bench.py.

The reference's test data (real VPU networks from S3, tests/download_test_data.sh:27-41) is absent, so the
BASELINE configs run on synthetic stand-ins.  Everything here is counter-based
(splitmix64 of seed and index), so any slice of any array can be regenerated without storing it.

Network model: Shreve's "topologically random" channel network -- a uniformly random full binary tree
(Remy's algorithm) -- which has ~50 % headwaters and a longest flow path ~ 2*sqrt(pi*n), i.e. the
in-degree-2, deep-and-narrow shape of real dendritic networks.  Reaches are then numbered in a random
topological order (upstream before downstream, tools.py:103-104 of the reference requires nothing more),
which is the least favourable ordering for any locality-based layout.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)

NETWORK_SEED = 20260320
PARAMS_SEED = 1
FORCING_SEED = 2


def splitmix64(x: np.ndarray | int) -> np.ndarray:
    """Finaliser of splitmix64 applied to (x + golden); vectorised over uint64 arrays."""
    with np.errstate(over='ignore'):
        z = (np.asarray(x, dtype=np.uint64) + _GOLD) & _M64
        z = ((z ^ (z >> np.uint64(30))) * _C1) & _M64
        z = ((z ^ (z >> np.uint64(27))) * _C2) & _M64
        return z ^ (z >> np.uint64(31))


def _seed_key(seed: int) -> np.uint64:
    return np.uint64((int(seed) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)


def u01(seed: int, index: np.ndarray | int) -> np.ndarray:
    """Uniform [0, 1) doubles from (seed, index): 53 high bits of splitmix64(seed * golden ^ index)."""
    key = _seed_key(seed) ^ np.asarray(index, dtype=np.uint64)
    return (splitmix64(key) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


@dataclass
class SynthNetwork:
    river_ids: np.ndarray  # int64[n], params-file order (topologically sorted)
    downstream_ids: np.ndarray  # int64[n], -1 at outlets
    down_index: np.ndarray  # int64[n] index of the downstream reach, -1 at outlets
    k: np.ndarray  # float64[n] seconds
    x: np.ndarray  # float64[n]

    @property
    def n(self) -> int:
        return int(self.river_ids.shape[0])

    def depth(self) -> int:
        """Number of reaches on the longest flow path."""
        d = np.zeros(self.n, dtype=np.int64)
        # params order is upstream-first, so walking it backwards visits downstream first
        for i in range(self.n - 1, -1, -1):
            j = self.down_index[i]
            if j >= 0:
                d[i] = d[j] + 1
        return int(d.max()) + 1 if self.n else 0


def _remy_parents(n_leaves: int, seed: int) -> np.ndarray:
    """Parent (= downstream) pointers of a uniformly random full binary tree with n_leaves leaves."""
    n_nodes = 2 * n_leaves - 1
    parent = [-1] * n_nodes
    # draw all picks at once: step k chooses uniformly among the 2k-1 existing nodes
    ks = np.arange(1, n_leaves, dtype=np.uint64)
    hi = splitmix64(_seed_key(seed) ^ ks) >> np.uint64(32)
    picks = ((hi * (np.uint64(2) * ks - np.uint64(1))) >> np.uint64(32)).astype(np.int64).tolist()
    for k, v in enumerate(picks, start=1):
        a = 2 * k - 1
        parent[a] = parent[v]
        parent[v] = a
        parent[a + 1] = a
    return np.asarray(parent, dtype=np.int64)


def _levels_from_headwaters(parent: np.ndarray) -> list[np.ndarray]:
    """Nodes grouped by longest distance from any headwater (level 0 = headwaters)."""
    n = parent.shape[0]
    indeg = np.bincount(parent[parent >= 0], minlength=n)
    remaining = indeg.copy()
    frontier = np.flatnonzero(indeg == 0)
    levels = []
    while frontier.size:
        levels.append(frontier)
        p = parent[frontier]
        p = p[p >= 0]
        if p.size == 0:
            break
        np.subtract.at(remaining, p, 1)
        cand = np.unique(p)
        frontier = cand[remaining[cand] == 0]
    return levels


def synth_network(n: int, seed: int = NETWORK_SEED, params_seed: int = PARAMS_SEED,
                  order: str = 'random') -> SynthNetwork:
    """
    n-reach random-topology network.  order: 'random' (random topological order, default),
    'bfs' (the engine's own order: farthest-from-outlet level first; needs no permutation pass on the GPU),
    'levels' (sorted by distance from the headwaters), 'postorder' (depth-first post-order: every reach right after the
    subtrees of its tributaries -- a common numbering of hydrography tables, and the one in which every sub-basin is a run of
    consecutive columns: the engine's direct row path applies).
    """
    if n < 1:
        raise ValueError('n must be >= 1')
    n_leaves = (n + 1) // 2
    parent = _remy_parents(n_leaves, seed)
    if parent.shape[0] < n:  # even n: one extra reach below the root
        root = int(np.flatnonzero(parent < 0)[0])
        parent = np.append(parent, -1)
        parent[root] = n - 1
    return _network_from_parents(parent, seed, params_seed, order)


def synth_network_chain(n: int, seed: int = NETWORK_SEED, p_chain: float = 0.0, n_outlets: int = 1, p_third: float = 0.0,
                        params_seed: int = PARAMS_SEED, order: str = 'random') -> SynthNetwork:
    """
    The generator SURVEY.md section 8(d) specifies, next to the Remy tree above: outlet-first growth.  Node 0 (and the other
    n_outlets - 1 first nodes) is an outlet with two open upstream slots; node u = 1, 2, ... attaches with probability
    p_chain to a slot of node u - 1 (it extends the current channel: this is the depth knob, long in-degree-1 runs as real
    networks have them) and otherwise to a uniformly random open slot; every new node opens two slots (three with probability
    p_third), so in-degree <= 2 (<= 3), about half the reaches of a p_chain = 0 network are headwaters, and several outlets
    make a forest.  Reaches are then numbered upstream-before-downstream (`order` as in synth_network).
    """
    if n < 1 or not (0.0 <= p_chain < 1.0) or n_outlets < 1:
        raise ValueError('need n >= 1, 0 <= p_chain < 1, n_outlets >= 1')
    n_outlets = min(n_outlets, n)
    idx = np.arange(n, dtype=np.int64)
    chain = (u01(seed + 11, idx) < p_chain).tolist()
    pick = u01(seed + 12, idx).tolist()
    third = (u01(seed + 13, idx) < p_third).tolist()
    parent = [-1] * n
    slots: list = []                # open upstream slots, one entry (the node) per slot
    for u in range(n):
        if u >= n_outlets:
            if chain[u] and u > n_outlets:      # the slots of node u - 1 are the last entries: nothing has touched them yet
                parent[u] = slots.pop()
            else:
                j = int(pick[u] * len(slots))
                parent[u] = slots[j]
                slots[j] = slots[-1]
                slots.pop()
        slots.append(u)
        slots.append(u)
        if third[u]:
            slots.append(u)
    return _network_from_parents(np.asarray(parent, dtype=np.int64), seed, params_seed, order)


def _postorder(parent: np.ndarray) -> np.ndarray:
    """Depth-first post-order of a forest given by parent pointers, as tools.postorder numbers it (rr_postorder: a reach's
    tributaries largest sub-basin first, outlets in ascending node number)."""
    from . import _lib
    down = np.ascontiguousarray(parent, dtype=np.int64)
    order = np.empty(down.shape[0], dtype=np.int64)
    _lib.check(_lib.lib().rr_postorder(down.shape[0], _lib.ptr(down), _lib.ptr(order)))
    return order


def _network_from_parents(parent: np.ndarray, seed: int, params_seed: int, order: str) -> SynthNetwork:
    n = parent.shape[0]
    levels = _levels_from_headwaters(parent)

    if order == 'random':
        key = np.zeros(n, dtype=np.float64)
        maxup = np.zeros(n, dtype=np.float64)
        for nodes in levels:
            key[nodes] = maxup[nodes] + u01(seed + 1, nodes) + 1e-9
            p = parent[nodes]
            m = p >= 0
            np.maximum.at(maxup, p[m], key[nodes][m])
        perm = np.argsort(key, kind='stable')
    elif order == 'levels':
        perm = np.concatenate(levels)
    elif order == 'bfs':
        # the engine's own order (rr_plan.hpp): farthest-from-outlet level first, each level in the order of
        # its downstream reaches.  A params file sorted this way needs no permutation pass on the GPU.
        from .engine import Plan
        from ._lib import RR_DEVICE_NONE
        first = np.concatenate(levels)
        rank = np.empty(n, dtype=np.int64)
        rank[first] = np.arange(n)
        pf = parent[first]
        down = np.where(pf >= 0, rank[np.maximum(pf, 0)], -1)
        has = down >= 0
        indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
        with Plan(indptr, down[has].astype(np.int32), device=RR_DEVICE_NONE) as plan:
            perm = first[plan.layout()[0].astype(np.int64)]
    elif order == 'postorder':
        perm = _postorder(parent)
    else:
        raise ValueError(f'unknown order {order!r}')

    index_of = np.empty(n, dtype=np.int64)
    index_of[perm] = np.arange(n, dtype=np.int64)
    pd = parent[perm]
    down_index = np.where(pd >= 0, index_of[np.maximum(pd, 0)], -1).astype(np.int64)
    assert np.all((down_index < 0) | (down_index > np.arange(n))), 'not topologically sorted'
    idx = np.arange(n, dtype=np.int64)
    river_ids = 1000 + 3 * idx + (idx % 3)
    downstream_ids = np.where(down_index >= 0, river_ids[np.maximum(down_index, 0)], -1).astype(np.int64)
    k = 900.0 + 6300.0 * u01(params_seed, idx)
    x = 0.05 + 0.40 * u01(params_seed + 7919, idx)
    return SynthNetwork(river_ids, downstream_ids, down_index, k, x)


def synth_qlateral(n: int, t0: int, t1: int, seed: int = FORCING_SEED, dt: float = 900.0) -> np.ndarray:
    """Runoff volumes (m^3 per dt) for steps [t0, t1): ql[t, i] = dt * u01(seed, t*n + i) -> (t1-t0, n) float64."""
    idx = (np.arange(t0, t1, dtype=np.uint64)[:, None] * np.uint64(n)) + np.arange(n, dtype=np.uint64)[None, :]
    return dt * u01(seed, idx)


def _i64(v: int) -> int:
    """The signed 64-bit integer with the bit pattern of the unsigned v."""
    v &= 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >= (1 << 63) else v


def u01_torch(seed: int, index):
    """u01 on a torch int64 tensor (any device), bit for bit: int64 products wrap like uint64 ones, logical right shifts are
    arithmetic ones with the sign bits masked off.  For the bench, whose forcing arrays take tens of seconds in numpy."""
    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)
    z = (index ^ _i64(int(_seed_key(seed)))) + _i64(int(_GOLD))
    z = (z ^ lsr(z, 30)) * _i64(int(_C1))
    z = (z ^ lsr(z, 27)) * _i64(int(_C2))
    z = z ^ lsr(z, 31)
    import torch
    return lsr(z, 11).to(torch.float64) * (1.0 / 9007199254740992.0)


def synth_qlateral_torch(n: int, t0: int, t1: int, device, columns=None, seed: int = FORCING_SEED, dt: float = 900.0, block: int = 32):
    """synth_qlateral computed on `device` (a torch tensor, the same bits); `columns`: only these reach indices."""
    import torch
    cols = torch.arange(n, dtype=torch.int64, device=device) if columns is None else torch.as_tensor(np.asarray(columns, dtype=np.int64), device=device)
    out = torch.empty((t1 - t0, cols.numel()), dtype=torch.float64, device=device)
    for r0 in range(t0, t1, block):
        r1 = min(t1, r0 + block)
        idx = torch.arange(r0, r1, dtype=torch.int64, device=device)[:, None] * n + cols[None, :]
        out[r0 - t0:r1 - t0] = dt * u01_torch(seed, idx)
    return out


def synth_runoff_depth(n: int, t0: int, t1: int, seed: int = FORCING_SEED) -> np.ndarray:
    """Runoff depths (m) in [0, 1e-3) for UnitMuskingum inputs."""
    idx = (np.arange(t0, t1, dtype=np.uint64)[:, None] * np.uint64(n)) + np.arange(n, dtype=np.uint64)[None, :]
    return 1e-3 * u01(seed + 104729, idx)


def synth_uh_kernel(n: int, n_ks: int, tr: float = 900.0, seed: int = PARAMS_SEED) -> np.ndarray:
    """
    (n_ks, n) unit-hydrograph kernel: a triangular pulse per basin, peak position and area seeded,
    normalised so that sum(kernel[:, j]) * tr == area[j] (the invariant of tests/test_uhkernels.py:19-30).
    """
    j = np.arange(n, dtype=np.int64)
    area = 1e6 + 4.9e7 * u01(seed + 15485863, j)
    peak = 1.0 + (n_ks - 2.0) * 0.5 * u01(seed + 32452843, j) if n_ks > 2 else np.full(n, 0.5)
    s = np.arange(n_ks, dtype=np.float64)[:, None] + 0.5
    base = np.maximum(n_ks * 1.0, 1.0)
    rise = s / peak[None, :]
    fall = (base - s) / np.maximum(base - peak[None, :], 1e-9)
    shape = np.maximum(np.minimum(rise, fall), 0.0)
    shape /= shape.sum(axis=0, keepdims=True)
    return shape * (area / tr)[None, :]

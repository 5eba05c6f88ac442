"""
Network structure helpers on the hot path (river_route/tools.py:75-109) plus the engine-order helper.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse

__all__ = ['adjacency_matrix', 'engine_order']


def adjacency_matrix(river_ids: np.ndarray, downstream_ids: np.ndarray) -> scipy.sparse.csc_matrix:
    """
    Sparse adjacency of the river network, A[downstream_idx, upstream_idx] = 1, as the reference builds it
    (river_route/tools.py:75-109): outlets (downstream id < 0) have no entry; raises ValueError for a downstream
    id that is not a river id and for inputs that are not sorted upstream -> downstream.  Vectorised (the
    reference walks a Python dict), same result, same first-failure-in-scan-order error.
    """
    rid = np.asarray(river_ids).astype(np.int64, copy=False).ravel()
    did = np.asarray(downstream_ids).astype(np.int64, copy=False).ravel()
    n = rid.shape[0]
    if did.shape[0] != n:
        raise ValueError('river_ids and downstream_ids must have the same length')
    up = np.flatnonzero(did >= 0)
    order = np.argsort(rid, kind='stable')
    sorted_ids = rid[order]
    # a repeated id resolves to its LAST index, like the reference's dict comprehension
    pos = np.searchsorted(sorted_ids, did[up], side='right') - 1
    known = (pos >= 0) & (sorted_ids[np.maximum(pos, 0)] == did[up])
    down = np.where(known, order[np.maximum(pos, 0)], -1)
    bad_unknown = ~known
    bad_order = known & (down <= up)
    if bad_unknown.any() or bad_order.any():
        first = int(np.flatnonzero(bad_unknown | bad_order)[0])
        if bad_unknown[first]:
            raise ValueError(f'Unknown downstream_river_id: {int(did[up[first]])}')
        raise ValueError('params_file must be topologically sorted upstream to downstream')
    data = np.ones(up.shape[0], dtype=np.float64)
    return scipy.sparse.csc_matrix((data, (down, up)), shape=(n, n))


def engine_order(river_ids: np.ndarray, downstream_ids: np.ndarray) -> np.ndarray:
    """
    Row order in which the HIP engine lays reaches out (farthest-from-outlet level first, each level in the
    order of its downstream reaches).  It is a valid topological order, so a params file (and its qlateral /
    state files) re-sorted with it is accepted by the reference unchanged and lets the engine skip its
    params-order <-> engine-order permutation passes.
    """
    from . import _lib
    from .engine import Plan
    A = adjacency_matrix(river_ids, downstream_ids)
    with Plan(A.indptr, A.indices, device=_lib.RR_DEVICE_NONE) as plan:
        return plan.layout()[0].astype(np.int64)

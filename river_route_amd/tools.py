"""
Network structure helpers: `adjacency_matrix` on the hot path (river_route/tools.py:75-109), the engine-order helper,
and the reference's offline network-editing utilities (river_route/tools.py:20-72) on plain arrays.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse

__all__ = ['adjacency_matrix', 'engine_order', 'postorder', 'connectivity_to_digraph', 'upstream_of', 'subset_configs_to_river']


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
    Row order of the streaming kernel's lag-ordered layout (farthest-from-outlet level first, each level in the order of
    its downstream reaches).  It is a valid topological order, so a params file (and its qlateral / state files)
    re-sorted with it is accepted by the reference unchanged; the streaming kernel then reads the caller's arrays in
    place.  (The time-tiled kernel, which routes all but very short calls, always goes through its record passes.)
    """
    from . import _lib
    from .engine import Plan
    A = adjacency_matrix(river_ids, downstream_ids)
    with Plan(A.indptr, A.indices, device=_lib.RR_DEVICE_NONE) as plan:
        return plan.layout()[0].astype(np.int64)


def postorder(river_ids: np.ndarray, downstream_ids: np.ndarray) -> np.ndarray:
    """
    Row order that sorts a network table into depth-first post-order: `order[k]` is the row that comes k-th, every reach right after
    the sub-basins of its tributaries (the small ones -- up to 256 reaches -- first, the main stem last, so that a stem's reaches are
    neighbours in the table: include/rr_hip.h, rr_postorder).  The rows may come in ANY order -- unlike `adjacency_matrix` this needs no
    topological sort, it makes one.  A params table (and the columns of its qlateral / state files) re-sorted with it,
    `table.iloc[order]`, is still valid for the reference (upstream before downstream, river_route/tools.py:103-104), and every
    sub-basin becomes a run of consecutive rows: the engine then routes RapidMuskingum straight from and to the (time, river) rows,
    without its record ring and permutation passes (the direct row path, `Plan.direct_info()`).
    Raises ValueError for a downstream id that is not a river id (as `adjacency_matrix` does) and for a network with a cycle.
    """
    from . import _lib
    rid = np.asarray(river_ids).astype(np.int64, copy=False).ravel()
    did = np.asarray(downstream_ids).astype(np.int64, copy=False).ravel()
    n = rid.shape[0]
    if did.shape[0] != n:
        raise ValueError('river_ids and downstream_ids must have the same length')
    by_id = np.argsort(rid, kind='stable')
    sorted_ids = rid[by_id]
    has = did >= 0
    pos = np.searchsorted(sorted_ids, did[has], side='right') - 1
    known = (pos >= 0) & (sorted_ids[np.maximum(pos, 0)] == did[has])
    if not known.all():
        raise ValueError(f'Unknown downstream_river_id: {int(did[has][np.flatnonzero(~known)[0]])}')
    down = np.full(n, -1, dtype=np.int64)
    down[has] = by_id[pos]
    order = np.empty(n, dtype=np.int64)
    try:
        _lib.check(_lib.lib().rr_postorder(n, _lib.ptr(down), _lib.ptr(order)))
    except _lib.RRError as e:
        raise ValueError(f'the river network is not a forest: {e.message}') from e
    return order


def connectivity_to_digraph(river_ids: np.ndarray, downstream_ids: np.ndarray):
    """networkx.DiGraph with one edge per reach, river -> downstream river, the -1 outlet sentinel included as a node
    (river_route/tools.py:57-72).  networkx is imported here: nothing on the routing path needs it."""
    import networkx as nx
    graph = nx.DiGraph()
    graph.add_edges_from(zip(np.asarray(river_ids).tolist(), np.asarray(downstream_ids).tolist()))
    return graph


def upstream_of(target_river: int, river_ids: np.ndarray, downstream_ids: np.ndarray) -> np.ndarray:
    """Boolean mask of `target_river` and every reach that drains into it, for ids in ANY order: a breadth-first walk up
    the network over a sorted edge list (what the reference gets from networkx.ancestors, river_route/tools.py:42-44)."""
    rid = np.asarray(river_ids).astype(np.int64, copy=False).ravel()
    did = np.asarray(downstream_ids).astype(np.int64, copy=False).ravel()
    if not (rid == target_river).any():
        raise ValueError(f'river {target_river} is not in river_ids')
    by_down = np.argsort(did, kind='stable')
    keys = did[by_down]
    keep = np.zeros(rid.shape[0], dtype=bool)
    frontier = np.flatnonzero(rid == target_river)
    while frontier.size:
        keep[frontier] = True
        lo, hi = np.searchsorted(keys, rid[frontier], 'left'), np.searchsorted(keys, rid[frontier], 'right')
        nxt = np.concatenate([by_down[a:b] for a, b in zip(lo, hi)]) if frontier.size else frontier
        frontier = nxt[~keep[nxt]]
    return keep


def subset_configs_to_river(target_river: int, params, out_params, weights=None, out_weights=None) -> None:
    """Routing parameters (and, when both weight paths are given, the grid weight table) of `target_river` and everything
    upstream of it; the target becomes the outlet of the subset, downstream_river_id = -1 (river_route/tools.py:20-54).
    Row order is kept, so a topologically sorted file stays sorted."""
    import pandas as pd
    table = pd.read_parquet(params)
    keep = upstream_of(target_river, table['river_id'].to_numpy(), table['downstream_river_id'].to_numpy())
    subset = table.loc[keep].copy()
    subset.loc[subset['river_id'] == target_river, 'downstream_river_id'] = -1
    subset.to_parquet(out_params)
    if weights is None or out_weights is None:
        return
    kept_ids = subset['river_id'].to_numpy()
    try:
        import xarray as xr
        with xr.open_dataset(weights) as ds:
            ds.isel(index=np.isin(ds['river_id'].values, kept_ids)).to_netcdf(out_weights)
        return
    except ImportError:
        pass
    from scipy.io import netcdf_file      # NetCDF-3 stand-in where xarray is absent: every variable along `index` is filtered
    with netcdf_file(str(weights), 'r', mmap=False) as src, netcdf_file(str(out_weights), 'w', version=2) as dst:
        mask = np.isin(np.array(src.variables['river_id'][:]), kept_ids)
        for name, size in src.dimensions.items():
            dst.createDimension(name, int(mask.sum()) if name == 'index' else size)
        for name, var in src.variables.items():
            data = np.array(var[:])
            if 'index' in var.dimensions:
                data = np.compress(mask, data, axis=var.dimensions.index('index'))
            out = dst.createVariable(name, data.dtype.newbyteorder('='), var.dimensions)
            out[:] = data
            for k, v in var._attributes.items():
                setattr(out, k, v)

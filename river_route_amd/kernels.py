"""
Drop-in replacements for the reference's kernel boundary: the three functions of
river_route/routers/_numba_kernels.py, argument for argument (numpy arrays in, state and discharge mutated in
place, None returned), executed by the HIP engine.  A maintainer of the reference swaps

    from ._numba_kernels import rapid_route      ->      from river_route_amd.kernels import rapid_route

(see INTEGRATION.md).  Plans are cached per network structure, like the reference's signature cache
(river_route/routers/TransformMuskingum.py:75-77).  No CPU fallback: without the GPU these raise.
"""
from __future__ import annotations

import hashlib

import numpy as np

from .engine import Plan

__all__ = ['muskingum_route', 'rapid_route', 'unit_route', 'clear_plan_cache', 'DEFAULT_DEVICE']

DEFAULT_DEVICE = 0
_PLANS: dict[tuple, Plan] = {}
_MAX_PLANS = 8


def _structure_key(indptr: np.ndarray, indices: np.ndarray, device: int) -> tuple:
    h = hashlib.blake2b(digest_size=16)
    h.update(np.ascontiguousarray(indptr, dtype=np.int32).tobytes())
    h.update(np.ascontiguousarray(indices, dtype=np.int32).tobytes())
    return (int(len(indptr)), int(len(indices)), h.hexdigest(), device)


def _plan_for(indptr, indices, device: int) -> Plan:
    key = _structure_key(indptr, indices, device)
    plan = _PLANS.get(key)
    if plan is None:
        if len(_PLANS) >= _MAX_PLANS:
            _PLANS.pop(next(iter(_PLANS))).close()
        plan = _PLANS[key] = Plan(indptr, indices, device)
    return plan


def clear_plan_cache() -> None:
    while _PLANS:
        _PLANS.popitem()[1].close()


def muskingum_route(csc_indptr, csc_indices, lhs_off_data, c2, c3, q_t, discharge_array,
                    num_output_steps, num_routing_per_output) -> None:
    """river_route/routers/_numba_kernels.py:9-46."""
    plan = _plan_for(csc_indptr, csc_indices, DEFAULT_DEVICE)
    plan.set_coeffs(lhs_off_data, c2, c3, None)
    plan.muskingum_route(q_t, discharge_array, int(num_output_steps), int(num_routing_per_output))


def rapid_route(csc_indptr, csc_indices, lhs_off_data, c2, c3, c4_dt, q_t, qlateral, discharge_array,
                num_substeps) -> None:
    """river_route/routers/_numba_kernels.py:50-84."""
    plan = _plan_for(csc_indptr, csc_indices, DEFAULT_DEVICE)
    plan.set_coeffs(lhs_off_data, c2, c3, c4_dt)
    plan.rapid_route(q_t, qlateral, discharge_array, int(num_substeps))


def full_structure_from_split(lhs_indptr, lhs_indices, a_hw_indptr, a_hw_indices, hw_idx, inner_idx, n_total):
    """Rebuild the CSC of the full adjacency from the inner x inner and inner x headwater blocks that
    UnitMuskingum._hook_before_route slices out of it (river_route/routers/UnitMuskingum.py:45-46)."""
    inner_idx = np.asarray(inner_idx, dtype=np.int64)
    hw_idx = np.asarray(hw_idx, dtype=np.int64)
    down = np.full(n_total, -1, dtype=np.int64)
    for indptr, indices, cols in ((lhs_indptr, lhs_indices, inner_idx), (a_hw_indptr, a_hw_indices, hw_idx)):
        indptr = np.asarray(indptr, dtype=np.int64)
        cnt = np.diff(indptr)
        if cnt.size and cnt.max(initial=0) > 1:
            raise NotImplementedError('unit_route: a reach with more than one downstream reach is not supported')
        has = cnt == 1
        down[cols[has]] = inner_idx[np.asarray(indices, dtype=np.int64)[indptr[:-1][has]]]
    has = down >= 0
    full_indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
    full_indices = down[has].astype(np.int32)
    return full_indptr, full_indices


def unit_route(lhs_indptr, lhs_indices, lhs_off_data,
               a_inner_indptr, a_inner_indices, a_inner_data,
               a_hw_indptr, a_hw_indices, a_hw_data,
               c1_inner, c2_inner, c3_inner,
               hw_idx, inner_idx,
               q_ch, q_full,
               convolved_lateral, discharge_array,
               num_substeps) -> None:
    """river_route/routers/_numba_kernels.py:89-171."""
    n_total = convolved_lateral.shape[1]
    inner_idx = np.asarray(inner_idx, dtype=np.int64)
    hw_idx = np.asarray(hw_idx, dtype=np.int64)
    indptr, indices = full_structure_from_split(lhs_indptr, lhs_indices, a_hw_indptr, a_hw_indices,
                                                hw_idx, inner_idx, n_total)
    plan = _plan_for(indptr, indices, DEFAULT_DEVICE)
    c1 = np.zeros(n_total)
    c2 = np.zeros(n_total)
    c3 = np.zeros(n_total)
    c1[inner_idx], c2[inner_idx], c3[inner_idx] = c1_inner, c2_inner, c3_inner
    lhs_off_data = np.asarray(lhs_off_data, dtype=np.float64)
    unit_weights = (np.all(np.asarray(a_inner_data) == 1.0) and np.all(np.asarray(a_hw_data) == 1.0)
                    and np.array_equal(lhs_off_data, -np.asarray(c1_inner)[np.asarray(lhs_indices)]))
    if unit_weights:      # what UnitMuskingum passes (UnitMuskingum.py:45-70): the time-tiled kernel
        plan.set_unit_weights(None, None)
        plan.set_coeffs(-c1[indices], c2, c3, None)
    else:
        # general edge data, per entry of the full CSC structure (one entry per reach that has a downstream reach, in reach
        # order): the streaming kernel multiplies and subtracts as _numba_kernels.py:126-139, 159-162 do
        has_down = np.diff(indptr) == 1
        entry_of = np.cumsum(has_down) - 1                                  # CSC entry of a reach's downstream edge
        a_full, lhs_full = np.zeros(indices.size), np.zeros(indices.size)
        for cols, ptr_, data, dst in ((inner_idx, a_inner_indptr, a_inner_data, a_full), (hw_idx, a_hw_indptr, a_hw_data, a_full),
                                      (inner_idx, lhs_indptr, lhs_off_data, lhs_full)):
            ptr_ = np.asarray(ptr_, dtype=np.int64)
            has = np.diff(ptr_) == 1
            dst[entry_of[cols[has]]] = np.asarray(data, dtype=np.float64)[ptr_[:-1][has]]
        plan.set_coeffs(lhs_full, c2, c3, None)
        plan.set_unit_weights(c1, a_full)
    plan.unit_route(q_ch, q_full, convolved_lateral, discharge_array, int(num_substeps))

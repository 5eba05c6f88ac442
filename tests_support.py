"""Shared by bench.py's cpu_baseline leg and tests: the headwater/inner split of UnitMuskingum._hook_before_route
(river_route/routers/UnitMuskingum.py:40-54) as plain arrays, needed to call the oracle's unit_route."""
import numpy as np
import scipy.sparse


def unit_split_arrays(indptr, indices, n):
    A = scipy.sparse.csc_matrix((np.ones(len(indices)), indices, indptr), shape=(n, n))
    incoming = np.asarray(A.sum(axis=1)).flatten()
    hw_idx = np.where(incoming == 0)[0]
    inner_idx = np.where(incoming != 0)[0]
    A_in = A[np.ix_(inner_idx, inner_idx)].tocsc()
    A_hw = A[np.ix_(inner_idx, hw_idx)].tocsc()
    return hw_idx, inner_idx, A_in, A_hw

"""
ctypes front-end of the CPU oracle (oracle/rr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under river_route_amd/ may import this module.

Parity: every function here is pinned to outputs of the reference itself (tests/golden/*.npz; tests/test_oracle.py,
tests/test_runoff.py).

The function signatures mirror the reference call sites so a parity test reads like the reference:
river_route/routers/_numba_kernels.py:9-14, 50-55, 89-99; uhkernels/UnitHydrograph.py:64-107;
routers/Muskingum.py:172-193; tools.py:75-109.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_i32p = np.ctypeslib.ndpointer(np.int32, flags='C_CONTIGUOUS')
_i64p = np.ctypeslib.ndpointer(np.int64, flags='C_CONTIGUOUS')
_f64p = np.ctypeslib.ndpointer(np.float64, flags='C_CONTIGUOUS')
_i64 = C.c_int64

_ERRORS = {
    -1: MemoryError('oracle: allocation failed'),
    -2: ValueError('Muskingum coefficients do not sum to 1, check routing parameters and time step'),
    -3: ValueError('Unknown downstream_river_id'),
    -4: ValueError('params_file must be topologically sorted upstream to downstream'),
}


def build(fast: bool = False, out_dir: str | None = None) -> str:
    """Compile the oracle with gcc if its shared object is missing or stale; returns the .so path."""
    name = 'librr_oracle_fast.so' if fast else 'librr_oracle.so'
    src = os.path.join(_HERE, 'rr_oracle.c')
    out = os.path.join(out_dir or _HERE, name)
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    flags = ['-O3', '-march=native', '-ffast-math'] if fast else ['-O2', '-fno-fast-math', '-ffp-contract=off']
    subprocess.check_call(['gcc', '-std=c11', *flags, '-fPIC', '-shared', '-o', out, src, '-lm'])
    return out


def _bind(lib: C.CDLL) -> C.CDLL:
    lib.orc_muskingum_route.argtypes = [_i64, _i32p, _i32p, _f64p, _f64p, _f64p, _f64p, _f64p, _i64, _i64]
    lib.orc_rapid_route.argtypes = [_i64, _i32p, _i32p, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _i64, _i64]
    lib.orc_unit_route.argtypes = [_i64, _i64, _i64, _i32p, _i32p, _f64p, _i32p, _i32p, _f64p, _i32p, _i32p, _f64p,
                                   _f64p, _f64p, _f64p, _i64p, _i64p, _f64p, _f64p, _f64p, _f64p, _i64, _i64]
    lib.orc_muskingum_coefficients.argtypes = [_i64, _f64p, _f64p, C.c_double, _f64p, _f64p, _f64p]
    lib.orc_adjacency_matrix.argtypes = [_i64, _i64p, _i64p, _i32p, _i32p]
    lib.orc_uh_convolve_incrementally.argtypes = [_i64, _i64, _f64p, _f64p, _f64p, _f64p]
    lib.orc_uh_convolve_incrementally.restype = None
    lib.orc_uh_convolve.argtypes = [_i64, _i64, _i64, _f64p, _f64p, _f64p, _f64p]
    for f in ('orc_muskingum_route', 'orc_rapid_route', 'orc_unit_route', 'orc_muskingum_coefficients',
              'orc_adjacency_matrix', 'orc_uh_convolve'):
        getattr(lib, f).restype = C.c_int
    return lib


_LIBS: dict[str, C.CDLL] = {}


def lib(fast: bool = False, out_dir: str | None = None) -> C.CDLL:
    path = build(fast, out_dir)
    if path not in _LIBS:
        _LIBS[path] = _bind(C.CDLL(path))
    return _LIBS[path]


def _check(rc: int) -> None:
    if rc != 0:
        raise _ERRORS.get(rc, RuntimeError(f'oracle error {rc}'))


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# ---- the three routing kernels, argument-for-argument as in _numba_kernels.py (mutating in place) ----

def muskingum_route(csc_indptr, csc_indices, lhs_off_data, c2, c3, q_t, discharge_array,
                    num_output_steps, num_routing_per_output, *, fast=False, out_dir=None):
    n = q_t.shape[0]
    assert discharge_array.shape == (num_output_steps, n)
    _check(lib(fast, out_dir).orc_muskingum_route(n, _i32(csc_indptr), _i32(csc_indices), _f64(lhs_off_data),
                                                  _f64(c2), _f64(c3), q_t, discharge_array,
                                                  num_output_steps, num_routing_per_output))


def rapid_route(csc_indptr, csc_indices, lhs_off_data, c2, c3, c4_dt, q_t, qlateral, discharge_array,
                num_substeps, *, fast=False, out_dir=None):
    n = q_t.shape[0]
    T = qlateral.shape[0]
    assert qlateral.shape == (T, n) and discharge_array.shape == (T, n)
    _check(lib(fast, out_dir).orc_rapid_route(n, _i32(csc_indptr), _i32(csc_indices), _f64(lhs_off_data),
                                              _f64(c2), _f64(c3), _f64(c4_dt), q_t, _f64(qlateral),
                                              discharge_array, T, num_substeps))


def unit_route(lhs_indptr, lhs_indices, lhs_off_data, a_inner_indptr, a_inner_indices, a_inner_data,
               a_hw_indptr, a_hw_indices, a_hw_data, c1_inner, c2_inner, c3_inner, hw_idx, inner_idx,
               q_ch, q_full, convolved_lateral, discharge_array, num_substeps, *, fast=False, out_dir=None):
    n_in, n_hw = len(inner_idx), len(hw_idx)
    T, n_total = convolved_lateral.shape
    assert discharge_array.shape == (T, n_total)
    _check(lib(fast, out_dir).orc_unit_route(
        n_in, n_hw, n_total, _i32(lhs_indptr), _i32(lhs_indices), _f64(lhs_off_data),
        _i32(a_inner_indptr), _i32(a_inner_indices), _f64(a_inner_data),
        _i32(a_hw_indptr), _i32(a_hw_indices), _f64(a_hw_data),
        _f64(c1_inner), _f64(c2_inner), _f64(c3_inner),
        np.ascontiguousarray(hw_idx, dtype=np.int64), np.ascontiguousarray(inner_idx, dtype=np.int64),
        q_ch, q_full, _f64(convolved_lateral), discharge_array, T, num_substeps))


# ---- host prep on the path ----

def muskingum_coefficients(k, x, dt_routing):
    """Muskingum.py:172-185 -> (c1, c2, c3); raises ValueError like the reference when they do not sum to 1."""
    k, x = _f64(k), _f64(x)
    c1, c2, c3 = (np.empty_like(k) for _ in range(3))
    _check(lib().orc_muskingum_coefficients(k.shape[0], k, x, float(dt_routing), c1, c2, c3))
    return c1, c2, c3


def adjacency_csc(river_ids, downstream_ids):
    """tools.py:75-109 -> (indptr int32[n+1], indices int32[nnz]) of A[down, up] = 1 in CSC."""
    rid = np.ascontiguousarray(river_ids, dtype=np.int64)
    did = np.ascontiguousarray(downstream_ids, dtype=np.int64)
    n = rid.shape[0]
    indptr = np.zeros(n + 1, dtype=np.int32)
    indices = np.zeros(max(n, 1), dtype=np.int32)
    _check(lib().orc_adjacency_matrix(n, rid, did, indptr, indices))
    return indptr, indices[:indptr[-1]].copy()


class UnitHydrograph:
    """UnitHydrograph.py:13-107 with the kernel handed over as a dense (n_ks, n) array."""

    def __init__(self, kernel):
        self.kernel = _f64(kernel)
        if self.kernel.ndim != 2:
            raise ValueError('kernel must be a 2D array')
        self.state = np.zeros_like(self.kernel)

    def convolve_incrementally(self, runoff_vector):
        n_ks, n = self.kernel.shape
        out = np.empty(n)
        lib().orc_uh_convolve_incrementally(n_ks, n, self.kernel, self.state, _f64(runoff_vector), out)
        return out

    def convolve(self, lateral):
        lateral = _f64(lateral)
        T = lateral.shape[0]
        n_ks, n = self.kernel.shape
        out = np.empty((T, n))
        _check(lib().orc_uh_convolve(T, n_ks, n, self.kernel, self.state, lateral, out))
        return out


def runoff_to_qlateral_core(weights, runoff_raw, catchment_area=None, cumulative=False, force_positive_runoff=False,
                            keep_nan=False):
    """The arithmetic of river_route/runoff.py:296-330 on arrays already extracted from the files: `weights` is the
    scipy CSR (n_rivers, n_points) of proportion * unit conversion (runoff.py:288-291), `runoff_raw` the (T, n_points)
    block of the runoff variable (float32 or float64).  numpy/scipy restatement, same statements in the same order:
    sparse product (296), cumulative -> incremental from the last row down (307-309), clip (310-311), NaN -> 0
    (327-329), times catchment area for volumes (331-332).  `keep_nan` stops before the fill, where the reference's
    irregular-time-step branch resamples (313-325).

    Pinned to the reference: tests/golden/runoff.npz holds what river_route.runoff.runoff_to_qlateral itself returned
    for five seeded cases (tests/golden/make_golden_runoff.py runs the reference function as written; xarray, absent
    from the image, is replaced there by a file-access stand-in that does no arithmetic).  tests/test_runoff.py checks
    this restatement -- through the host logic of river_route_amd.runoff -- against those outputs, and against an
    independent dense evaluation."""
    q = np.array(np.asarray(weights @ np.asarray(runoff_raw).T).T, dtype=np.float64)      # (time, n_rivers)
    if cumulative:
        for i in range(q.shape[0] - 1, 0, -1):
            q[i] -= q[i - 1]
    if force_positive_runoff:
        np.clip(q, 0, None, out=q)
    if not keep_nan:
        mask = np.isnan(q)
        if mask.any():
            q[mask] = 0.0
    if catchment_area is not None:
        q *= np.asarray(catchment_area, dtype=np.float64)[np.newaxis, :]
    return q

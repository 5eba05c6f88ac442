"""
`UnitHydrograph`: the reference's stateful runoff transformer (river_route/uhkernels/UnitHydrograph.py:13-107)
with the convolution executed by the HIP engine (rr_uh_convolve).  Same constructor, attributes
(`kernel`, `state`, both (n_kernel_steps, n_basins) float64), state file layout and carry-over semantics.
"""
from __future__ import annotations

import numpy as np

from .engine import uh_convolve

__all__ = ['UnitHydrograph']


class UnitHydrograph:
    kernel: np.ndarray
    state: np.ndarray

    def __init__(self, kernel_file, device: int = 0) -> None:
        import scipy.sparse
        self.kernel = np.ascontiguousarray(scipy.sparse.load_npz(kernel_file).toarray().astype(np.float64, copy=False))
        if self.kernel.ndim != 2:
            raise ValueError('kernel must be a 2D array')
        self.device = device
        self.reset_state()

    @classmethod
    def from_array(cls, kernel: np.ndarray, device: int = 0) -> 'UnitHydrograph':
        """Build from a dense (n_kernel_steps, n_basins) array without a file (tests, benchmarks)."""
        self = cls.__new__(cls)
        self.kernel = np.ascontiguousarray(kernel, dtype=np.float64)
        if self.kernel.ndim != 2:
            raise ValueError('kernel must be a 2D array')
        self.device = device
        self.reset_state()
        return self

    def reset_state(self) -> None:
        self.state = np.zeros_like(self.kernel, dtype=np.float64)

    def set_state(self, path) -> 'UnitHydrograph':
        """Carry-over state from parquet; the file holds basins as rows: (n_basins, n_kernel_steps)."""
        import pandas as pd
        st = pd.read_parquet(path).T.to_numpy(dtype=np.float64, copy=True)
        if st.shape != self.kernel.shape:
            raise ValueError(f'state shape {st.shape} does not match kernel shape {self.kernel.shape}')
        self.state = np.ascontiguousarray(st)
        return self

    def write_state(self, path) -> None:
        import pandas as pd
        pd.DataFrame(self.state.T).to_parquet(path)

    def convolve(self, lateral: np.ndarray) -> np.ndarray:
        """(t, n_basins) runoff depths -> (t, n_basins) lateral inflow; carries and updates `state`."""
        if not (self.state.flags['C_CONTIGUOUS'] and self.state.flags['WRITEABLE'] and self.state.dtype == np.float64):
            self.state = np.array(self.state, dtype=np.float64, order='C')
        return uh_convolve(self.kernel, self.state, lateral, device=self.device)

    def convolve_incrementally(self, runoff_vector: np.ndarray) -> np.ndarray:
        """One time step (UnitHydrograph.py:64-75): identical to convolving a single row with the carried state."""
        return self.convolve(np.asarray(runoff_vector, dtype=np.float64)[None, :])[0]


# ---------------------------------------------------------------------------------------------------------------
# SCS unit-hydrograph kernel builders (river_route/uhkernels/_SCSBase.py:36-81).  Offline pre-processing that
# produces the (n_steps, n_basins) kernel file UnitMuskingum consumes; plain numpy, no GPU work.
# Dimensionless hydrograph tables: NRCS National Engineering Handbook Part 630, Chapter 16, Table 16-1.
# ---------------------------------------------------------------------------------------------------------------

def _scs_kernel(table_t: np.ndarray, table_q: np.ndarray, tc, area, tr: float) -> np.ndarray:
    """
    Kernel[s, j] = mean flow (m^2/s per metre of runoff) of basin j during [s*tr, (s+1)*tr] after a unit depth of
    runoff generated over `tr` seconds.  The dimensionless hydrograph (t/tp, q/qp) is integrated by trapezoids to a
    cumulative curve; lag 0.6*tc, time to peak lag + tr/2, base = last table time * tp, peak flow chosen so the
    hydrograph volume equals the basin area; each kernel row is the increase of the cumulative curve over one
    interval divided by tr, so sum(kernel[:, j]) * tr == area[j].
    """
    if float(tr) <= 0:
        raise ValueError('tr must be > 0')
    tc = np.asarray(tc, dtype=np.float64)
    area = np.asarray(area, dtype=np.float64)
    if tc.ndim != 1:
        raise ValueError('tc must be a 1D float array')
    if area.ndim != 1:
        raise ValueError('area must be a 1D float array')
    if tc.shape != area.shape:
        raise ValueError('tc and area must have the same length')
    tr = float(tr)
    cumulative = np.zeros(table_t.shape[0])
    cumulative[1:] = np.cumsum(0.5 * (table_q[1:] + table_q[:-1]) * np.diff(table_t))
    unit_area = cumulative[-1]
    t_peak = 0.6 * tc + 0.5 * tr
    t_base = table_t[-1] * t_peak
    q_peak = area / (unit_area * t_peak)
    steps = int(np.ceil(t_base / tr).astype(int).max())
    edges = np.minimum(np.arange(steps + 1, dtype=np.int64)[:, None] * tr, t_base)        # (steps + 1, n_basins)
    frac = np.interp((edges / t_peak).ravel(), table_t, cumulative, right=unit_area).reshape(edges.shape)
    volume = frac * (q_peak * t_peak)
    return np.diff(volume, axis=0) / tr


class _SCSKernel:
    _table_t: np.ndarray
    _table_q: np.ndarray

    def __init__(self, *, tc, area, tr: float) -> None:
        self.kernel = _scs_kernel(self._table_t, self._table_q, tc, area, tr)
        self.tr = float(tr)
        self.tc = np.asarray(tc, dtype=np.float64)
        self.area = np.asarray(area, dtype=np.float64)
        self.tl = 0.6 * self.tc
        self.tp = self.tl + self.tr / 2.0
        self.tb = self._table_t[-1] * self.tp

    def save(self, path) -> None:
        """Write the kernel as the scipy sparse npz file `uh_kernel_file` points at."""
        import scipy.sparse
        scipy.sparse.save_npz(path, scipy.sparse.csr_matrix(self.kernel))


class SCSTriangular(_SCSKernel):
    """SCS triangular dimensionless unit hydrograph (river_route/uhkernels/SCSTriangular.py:8-25)."""
    _table_t = np.array([0.0, 1.0, 2.67])
    _table_q = np.array([0.0, 1.0, 0.0])


class SCSCurvilinear(_SCSKernel):
    """SCS curvilinear dimensionless unit hydrograph (river_route/uhkernels/SCSCurvilinear.py:8-36)."""
    _table_t = np.concatenate([np.arange(0.0, 2.01, 0.1), np.arange(2.2, 4.01, 0.2), [4.5, 5.0]]).round(10)
    _table_q = np.array([0.000, 0.015, 0.075, 0.160, 0.280, 0.430, 0.600, 0.770, 0.890, 0.970, 1.000,
                         0.980, 0.920, 0.840, 0.750, 0.660, 0.560, 0.460, 0.390, 0.330, 0.280,
                         0.207, 0.147, 0.107, 0.077, 0.055, 0.040, 0.029, 0.021, 0.015, 0.011,
                         0.005, 0.000])


__all__ += ['SCSTriangular', 'SCSCurvilinear']

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

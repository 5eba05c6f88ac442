"""
`UnitMuskingum` (river_route/routers/UnitMuskingum.py:11-104): unit-hydrograph lateral inflow superimposed on
Muskingum channel routing; headwaters carry the convolved runoff only.
"""
from __future__ import annotations

import numpy as np

from ..uhkernels import UnitHydrograph
from .transform import TransformMuskingum

__all__ = ['UnitMuskingum']


class UnitMuskingum(TransformMuskingum):
    _ROUTER_REQUIRED_CONFIGS = ('uh_kernel_file',)
    _uh: UnitHydrograph | None = None
    _as_volumes = False   # inputs are runoff depths (m)

    def _hook_before_route(self) -> None:
        if self._uh is None:
            self.logger.debug('Loading UH kernel')
            self._uh = UnitHydrograph(self.cfg.uh_kernel_file, device=self.cfg.device)
            if self.cfg.uh_state_init_file:
                self._uh.set_state(self.cfg.uh_state_init_file)
        if not hasattr(self, 'hw_idx'):
            incoming = np.asarray(self.A.sum(axis=1)).flatten()
            self.hw_idx = np.where(incoming == 0)[0]
            self.inner_idx = np.where(incoming != 0)[0]
            self.logger.info(
                f'Headwater split: {len(self.hw_idx)} headwater, {len(self.inner_idx)} inner '
                f'({len(self.hw_idx) / self.A.shape[0] * 100:.0f}% excluded from solve)')

    def _check_kernel(self) -> None:
        """The kernel file must describe this network: one column of taps per reach (scipy's fftconvolve raises in the
        reference; raw device addresses would not)."""
        n = self.A.shape[0]
        if self._uh.kernel.ndim != 2 or self._uh.kernel.shape[1] != n or np.shape(self._uh.state) != self._uh.kernel.shape:
            raise ValueError(f'unit-hydrograph kernel {self._uh.kernel.shape} / state {np.shape(self._uh.state)} do not match '
                             f'the network: expected (n_kernel_steps, {n})')

    def _seed(self) -> np.ndarray:
        """Channel and full discharge of the inner reaches both restart from channel_state at every file (UnitMuskingum.py:78-79)."""
        return np.array(self.channel_state[self.inner_idx], dtype=np.float64, order='C')

    def _router(self, qlateral: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        self._check_kernel()
        lateral = self._uh.convolve(qlateral)                       # rr_uh_convolve
        n = self.river_ids.shape[0]
        routed = np.zeros((self.num_runoff_steps, n), dtype=np.float64)
        q_ch = self._seed()
        q_full = q_ch.copy()
        self._upload_coefficients(None, ('unit',))
        self._plan.unit_route(q_ch, q_full, lateral, routed, self.num_routing_steps_per_runoff)
        state = np.empty(n, dtype=np.float64)
        state[self.hw_idx] = lateral[-1][self.hw_idx]       # a headwater's state is its last lateral inflow
        state[self.inner_idx] = q_full
        return state, routed

    _engine_router = _router

    def _router_device(self, qlateral: np.ndarray, rows_per_output: int) -> tuple[np.ndarray, np.ndarray]:
        from .._lib import RR_E_UNSUPPORTED, RRError
        from ._device import Arena
        self._check_kernel()
        depth = self._check_lateral(qlateral, keep_float32=True)
        if depth.dtype == np.float32:      # a float32 file: uploaded as it is, converted in the pass that convolves it into the engine's records
            try:
                with Arena(self.cfg.device) as arena:
                    return self._route_on_device_f32in(arena, arena.put(depth), depth.shape[0], rows_per_output)
            except RRError as e:
                if e.code != RR_E_UNSUPPORTED:
                    raise
            depth = depth.astype(np.float64)
        with Arena(self.cfg.device) as arena:
            return self._route_on_device(arena, arena.put(depth), depth.shape[0], rows_per_output)

    def _route_file_to_file(self, rows, dates_out, discharge_file, runoff_file):
        """One runoff-depth file whose float32 rows lie flat in the file (nc3.RowBlock) -> the discharge file without a host array of
        either: engine.rows_upload, rr_unit_route_uh_f32in_dev (convolution fused in, the file's byte order converted in the kernels),
        nc3.create_discharge_file, engine.rows_download.  Returns the router state (UnitMuskingum.py:95-97), or None where the fused
        form does not apply or the file does not fit the card -- the UH state is then as it was."""
        from .. import nc3
        from .._lib import RR_E_ALLOC, RR_E_UNSUPPORTED, RRError
        from ..engine import rows_download, rows_upload
        from ._device import Arena, DeviceOutOfMemory
        self._check_kernel()
        T, n, nsub, per = rows.rows, self.A.shape[0], self.num_routing_steps_per_runoff, self.num_runoff_steps_per_discharge
        if T != self.num_runoff_steps:
            raise ValueError(f'lateral inflow has shape ({T}, {rows.cols}), expected ({self.num_runoff_steps}, {n}) from the time options')
        n_ks = self._uh.kernel.shape[0]
        self._upload_coefficients(None, ('unit',))
        seed = self._seed()
        dev = self.cfg.device
        try:
            with Arena(dev) as arena:
                d_depth = arena.empty(T * n * 4)
                d_out = arena.empty((T // per) * n * 4)
                d_kern = arena.put(self._uh.kernel)
                d_state = arena.put(np.ascontiguousarray(self._uh.state, dtype=np.float64))
                d_qch, d_qfull, d_final = arena.put(seed), arena.put(seed), arena.empty(n * 8)
                rows_upload(d_depth, n * 4, rows.path, rows.offset, rows.pitch, n * 4, T, device=dev)
                self._plan.set_row_format(rows.big_endian, True)
                try:
                    self._plan.unit_route_uh_f32in_dev(d_qch, d_qfull, d_final, d_kern, d_state, n_ks, d_depth, T, nsub, discharge32=d_out, factor=per)
                finally:
                    self._plan.set_row_format(False, False)
                out = nc3.create_discharge_file(discharge_file, dates_out, self.river_ids, self.cfg.var_river_id, self.cfg.var_discharge, runoff_file)
                rows_download(d_out, n * 4, out.path, out.offset, out.pitch, n * 4, T // per, device=dev)
                state = d_final.download(np.float64, (n,))
                self._uh.state = d_state.download(np.float64, self._uh.kernel.shape)
                return state
        except DeviceOutOfMemory:
            return None
        except RRError as e:
            if e.code not in (RR_E_UNSUPPORTED, RR_E_ALLOC):
                raise
            return None

    def _route_on_device_f32in(self, arena, d_depth32, T: int, rows_per_output: int) -> tuple[np.ndarray, np.ndarray]:
        """Runoff depths on the device as (T, n) float32 rows -> (router state, float32 discharge rows): rr_unit_route_uh_f32in_dev,
        bit for bit what the float64 rows give (float32 -> float64 is exact).  RR_E_UNSUPPORTED (a call the time-tiled kernel does not
        take, more than 64 kernel steps) leaves the UH state as it was: the caller converts the rows and takes the float64 path."""
        self._check_kernel()
        n = self.A.shape[0]
        n_ks, nsub = self._uh.kernel.shape[0], self.num_routing_steps_per_runoff
        self._upload_coefficients(None, ('unit',))
        seed = self._seed()
        d_kern = arena.put(self._uh.kernel)
        d_state = arena.put(np.ascontiguousarray(self._uh.state, dtype=np.float64))
        d_qch, d_qfull, d_final = arena.put(seed), arena.put(seed), arena.empty(n * 8)
        d_f32 = arena.empty((T // rows_per_output) * n * 4)
        self._plan.unit_route_uh_f32in_dev(d_qch, d_qfull, d_final, d_kern, d_state, n_ks, d_depth32, T, nsub,
                                           discharge32=d_f32, factor=rows_per_output)
        q_array = d_f32.download(np.float32, (T // rows_per_output, n))
        state = d_final.download(np.float64, (n,))
        self._uh.state = d_state.download(np.float64, self._uh.kernel.shape)
        return state, q_array

    def _route_on_device(self, arena, d_depth, T: int, rows_per_output: int) -> tuple[np.ndarray, np.ndarray]:
        """Runoff depths already on the device, (T, n) float64 rows -> (router state, float32 discharge rows).  Where the
        engine takes it (time-tiled call, at most 64 kernel steps) the convolution is fused into the pass that turns rows
        into records (rr_unit_route_uh_dev): the convolved lateral never exists as rows; otherwise rr_uh_convolve_dev, then
        the routing call."""
        from .._lib import RR_E_UNSUPPORTED, RRError
        from ..engine import uh_convolve_dev
        from ._device import float32_rows
        self._check_kernel()
        n = self.A.shape[0]
        n_ks, nsub = self._uh.kernel.shape[0], self.num_routing_steps_per_runoff
        self._upload_coefficients(None, ('unit',))
        seed = self._seed()
        d_kern = arena.put(self._uh.kernel)
        d_state = arena.put(np.ascontiguousarray(self._uh.state, dtype=np.float64))
        d_qch, d_qfull, d_final = arena.put(seed), arena.put(seed), arena.empty(n * 8)
        d_f32 = arena.empty((T // rows_per_output) * n * 4)
        try:
            self._plan.unit_route_uh_dev(d_qch, d_qfull, d_final, d_kern, d_state, n_ks, d_depth, T, nsub,
                                         discharge32=d_f32, factor=rows_per_output)
            q_array = d_f32.download(np.float32, (T // rows_per_output, n))
            state = d_final.download(np.float64, (n,))
        except RRError as e:
            if e.code != RR_E_UNSUPPORTED:
                raise
            arena.release(d_f32)
            d_conv = arena.empty(T * n * 8)
            uh_convolve_dev(d_kern, d_state, d_depth, d_conv, T, n_ks, n, device=self.cfg.device)
            q_array = float32_rows(
                arena, T, n, rows_per_output,
                fused=lambda d32: self._plan.unit_route_f32_dev(d_qch, d_qfull, d_conv, T, d32, T, nsub, rows_per_output),
                plain=lambda d64: self._plan.unit_route_dev(d_qch, d_qfull, d_conv, T, d64, T, T, nsub))
            state = d_conv.download(np.float64, (n,), offset=(T - 1) * n * 8)    # headwaters keep the last lateral row
            state[self.inner_idx] = d_qfull.download(np.float64, seed.shape)
        self._uh.state = d_state.download(np.float64, self._uh.kernel.shape)
        return state, q_array

    def _write_final_state(self) -> None:
        super()._write_final_state()
        if self.cfg.uh_state_final_file and self._uh is not None:
            self._uh.write_state(self.cfg.uh_state_final_file)

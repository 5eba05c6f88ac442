"""`RapidMuskingum` (river_route/routers/RapidMuskingum.py:10-33): Muskingum routing with direct lateral inflow."""
from __future__ import annotations

import numpy as np

from .transform import TransformMuskingum

__all__ = ['RapidMuskingum']


class RapidMuskingum(TransformMuskingum):
    _as_volumes = True   # qlateral is a volume (m3) per runoff step

    def _lateral_coefficient(self) -> None:
        """c4 / dt_runoff turns a volume per runoff step into the discharge term of the routing step (RapidMuskingum.py:24)."""
        self._upload_coefficients(self.c4 / self.dt_runoff, ('rapid', int(self.dt_runoff)))

    def _router(self, qlateral: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        state = np.array(self.channel_state, dtype=np.float64, order='C')
        routed = np.zeros((self.num_runoff_steps, state.shape[0]), dtype=np.float64)
        self._lateral_coefficient()
        self._plan.rapid_route(state, qlateral, routed, self.num_routing_steps_per_runoff)
        return state, routed

    _engine_router = _router

    def _router_device(self, qlateral: np.ndarray, rows_per_output: int) -> tuple[np.ndarray, np.ndarray]:
        from .._lib import RR_E_UNSUPPORTED, RRError
        from ._device import Arena
        ql = self._check_lateral(qlateral, keep_float32=True)
        if ql.dtype == np.float32:      # a float32 file: uploaded as it is, converted in the pass that fills the engine's records
            try:
                with Arena(self.cfg.device) as arena:
                    return self._route_on_device_f32in(arena, arena.put(ql), ql.shape[0], rows_per_output)
            except RRError as e:
                if e.code != RR_E_UNSUPPORTED:
                    raise
            ql = ql.astype(np.float64)
        with Arena(self.cfg.device) as arena:
            return self._route_on_device(arena, arena.put(ql), ql.shape[0], rows_per_output)

    def _route_file_to_file(self, rows, dates_out, discharge_file, runoff_file):
        """One qlateral file whose float32 rows lie flat in the file (nc3.RowBlock) -> the discharge file, without the (time, river)
        block ever being a host array: engine.rows_upload (page cache -> pinned chunks -> device), rr_rapid_route_f32in_dev with the
        file's byte order converted in the kernels that read and write the rows (Plan.set_row_format), the discharge file's header
        (nc3.create_discharge_file: the layout of Muskingum.py:337-351), engine.rows_download.  Returns the final state, or None where
        the fused float32 form does not apply or the file does not fit the card (the caller then takes the file as an array)."""
        from .. import nc3
        from .._lib import RR_E_ALLOC, RR_E_UNSUPPORTED, RRError
        from ..engine import rows_download, rows_upload
        from ._device import Arena, DeviceOutOfMemory
        T, n, nsub, per = rows.rows, self.A.shape[0], self.num_routing_steps_per_runoff, self.num_runoff_steps_per_discharge
        if T != self.num_runoff_steps:
            raise ValueError(f'lateral inflow has shape ({T}, {rows.cols}), expected ({self.num_runoff_steps}, {n}) from the time options')
        self._lateral_coefficient()
        dev = self.cfg.device
        try:
            with Arena(dev) as arena:
                d_ql = arena.empty(T * n * 4)
                d_out = arena.empty((T // per) * n * 4)
                d_q = arena.put(np.array(self.channel_state, dtype=np.float64, order='C'))
                rows_upload(d_ql, n * 4, rows.path, rows.offset, rows.pitch, n * 4, T, device=dev)
                self._plan.set_row_format(rows.big_endian, True)      # a NetCDF-3 discharge file: big-endian rows out
                try:
                    self._plan.rapid_route_f32in_dev(d_q, d_ql, T, T, nsub, discharge32=d_out, factor=per)
                finally:
                    self._plan.set_row_format(False, False)
                out = nc3.create_discharge_file(discharge_file, dates_out, self.river_ids, self.cfg.var_river_id, self.cfg.var_discharge, runoff_file)
                rows_download(d_out, n * 4, out.path, out.offset, out.pitch, n * 4, T // per, device=dev)
                return d_q.download(np.float64, (n,))
        except DeviceOutOfMemory:
            return None
        except RRError as e:
            if e.code not in (RR_E_UNSUPPORTED, RR_E_ALLOC):
                raise
            return None

    def _route_on_device_f32in(self, arena, d_ql32, T: int, rows_per_output: int) -> tuple[np.ndarray, np.ndarray]:
        """Lateral volumes on the device as (T, n) float32 rows -> (final state, float32 discharge rows): rr_rapid_route_f32in_dev,
        bit for bit what the float64 rows give (float32 -> float64 is exact)."""
        from ._device import float32_rows
        n, nsub = self.A.shape[0], self.num_routing_steps_per_runoff
        self._lateral_coefficient()
        d_q = arena.put(np.array(self.channel_state, dtype=np.float64, order='C'))
        q_array = float32_rows(
            arena, T, n, rows_per_output,
            fused=lambda d32: self._plan.rapid_route_f32in_dev(d_q, d_ql32, T, T, nsub, discharge32=d32, factor=rows_per_output),
            plain=lambda d64: self._plan.rapid_route_f32in_dev(d_q, d_ql32, T, T, nsub, discharge=d64, out_rows=T))
        return d_q.download(np.float64, (n,)), q_array

    def _route_on_device(self, arena, d_ql, T: int, rows_per_output: int) -> tuple[np.ndarray, np.ndarray]:
        """Lateral volumes already on the device, (T, n) float64 rows -> (final state, float32 discharge rows)."""
        from ._device import float32_rows
        n, nsub = self.A.shape[0], self.num_routing_steps_per_runoff
        self._lateral_coefficient()
        d_q = arena.put(np.array(self.channel_state, dtype=np.float64, order='C'))
        q_array = float32_rows(
            arena, T, n, rows_per_output,
            fused=lambda d32: self._plan.rapid_route_f32_dev(d_q, d_ql, T, d32, T, nsub, rows_per_output),
            plain=lambda d64: self._plan.rapid_route_dev(d_q, d_ql, T, d64, T, T, nsub))
        return d_q.download(np.float64, (n,)), q_array

    def _router_device_runoff(self, source, rows_per_output: int) -> tuple[np.ndarray, np.ndarray]:
        """Gridded runoff of one file.  With one routing step per runoff step the catchment volumes are computed inside the
        pass that fills the engine's records (rr_rapid_route_runoff_dev): they never exist as (time, river) rows, on the host
        or in HBM (13.7 ms against 16.3 ms for 1M reaches x 744 steps, profiles/r02_runoff_path.txt).  Otherwise, or where the
        engine answers RR_E_UNSUPPORTED, rr_runoff_to_qlateral_dev first and the routing call on its device rows."""
        from .._lib import RR_E_UNSUPPORTED, RRError
        from ._device import Arena
        T, n = source.runoff_tp.shape[0], self.A.shape[0]
        if self.num_routing_steps_per_runoff != 1 or source.river_ids.shape[0] != n or T != self.num_runoff_steps:
            return super()._router_device_runoff(source, rows_per_output)
        block = source.point_major()
        self._lateral_coefficient()
        try:
            with Arena(self.cfg.device) as arena:
                d_q = arena.put(np.array(self.channel_state, dtype=np.float64, order='C'))
                d_f32 = arena.empty((T // rows_per_output) * n * 4)
                self._plan.rapid_route_runoff_dev(
                    d_q, block.shape[0], arena.put(source.indptr), arena.put(source.indices), arena.put(source.weights), arena.put(block),
                    block.dtype == np.float32, 1, block.shape[1], arena.put(source.area), source.flags, T,
                    discharge32=d_f32, factor=rows_per_output)
                return d_q.download(np.float64, (n,)), d_f32.download(np.float32, (T // rows_per_output, n))
        except RRError as e:
            if e.code != RR_E_UNSUPPORTED:
                raise
        return super()._router_device_runoff(source, rows_per_output)

"""`RapidMuskingum` (river_route/routers/RapidMuskingum.py:10-33): Muskingum routing with direct lateral inflow."""
from __future__ import annotations

import numpy as np

from .transform import TransformMuskingum

__all__ = ['RapidMuskingum']


class RapidMuskingum(TransformMuskingum):
    _as_volumes = True   # qlateral is a volume (m3) per runoff step

    def _router(self, qlateral: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        n = self.A.shape[0]
        discharge_array = np.zeros((self.num_runoff_steps, n), dtype=np.float64)
        q_t = np.array(self.channel_state, dtype=np.float64, order='C')
        c4_dt = self.c4 / self.dt_runoff
        self._upload_coefficients(c4_dt, ('rapid', int(self.dt_runoff)))
        self._plan.rapid_route(q_t, qlateral, discharge_array, self.num_routing_steps_per_runoff)
        return q_t, discharge_array

    def _router_device(self, qlateral: np.ndarray, rows_per_output: int) -> tuple[np.ndarray, np.ndarray]:
        from ..engine import DeviceBuffer, resample_cast_dev
        ql = self._check_lateral(qlateral)
        T, n = ql.shape
        dev = self.cfg.device
        self._upload_coefficients(self.c4 / self.dt_runoff, ('rapid', int(self.dt_runoff)))
        bufs = []
        try:
            d_ql = DeviceBuffer(ql.nbytes, dev).upload(ql); bufs.append(d_ql)
            d_out = DeviceBuffer(ql.nbytes, dev); bufs.append(d_out)
            d_q = DeviceBuffer(n * 8, dev).upload(np.array(self.channel_state, dtype=np.float64, order='C')); bufs.append(d_q)
            d_f32 = DeviceBuffer((T // rows_per_output) * n * 4, dev); bufs.append(d_f32)
            self._plan.rapid_route_dev(d_q, d_ql, T, d_out, T, T, self.num_routing_steps_per_runoff)
            resample_cast_dev(d_out, T, n, rows_per_output, d_f32, dev)
            q_array = d_f32.download(np.float32, (T // rows_per_output, n))
            q_t = d_q.download(np.float64, (n,))
        finally:
            for b in bufs:
                b.free()
        return q_t, q_array

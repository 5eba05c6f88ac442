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

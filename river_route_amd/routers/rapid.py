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
        from ._device import Arena
        ql = self._check_lateral(qlateral)
        with Arena(self.cfg.device) as arena:
            return self._route_on_device(arena, arena.put(ql), ql.shape[0], rows_per_output)

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

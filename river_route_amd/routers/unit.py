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

    def _router(self, qlateral: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        convolved = self._uh.convolve(qlateral)                       # rr_uh_convolve
        n = self.river_ids.shape[0]
        discharge_array = np.zeros((self.num_runoff_steps, n), dtype=np.float64)
        # both channel and full discharge are re-seeded from channel_state at every file (UnitMuskingum.py:78-79)
        q_ch = np.array(self.channel_state[self.inner_idx], dtype=np.float64, order='C')
        q_full = q_ch.copy()
        self._upload_coefficients(None, ('unit',))
        self._plan.unit_route(q_ch, q_full, convolved, discharge_array, self.num_routing_steps_per_runoff)
        q_final = np.empty(n, dtype=np.float64)
        q_final[self.hw_idx] = convolved[-1][self.hw_idx]
        q_final[self.inner_idx] = q_full
        return q_final, discharge_array

    def _write_final_state(self) -> None:
        super()._write_final_state()
        if self.cfg.uh_state_final_file and self._uh is not None:
            self._uh.write_state(self.cfg.uh_state_final_file)

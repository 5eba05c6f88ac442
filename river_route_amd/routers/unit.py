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

    def _router_device(self, qlateral: np.ndarray, rows_per_output: int) -> tuple[np.ndarray, np.ndarray]:
        from ..engine import DeviceBuffer, resample_cast_dev, uh_convolve_dev
        depth = self._check_lateral(qlateral)
        T, n = depth.shape
        dev = self.cfg.device
        n_ks = self._uh.kernel.shape[0]
        self._upload_coefficients(None, ('unit',))
        bufs = []
        try:
            d_depth = DeviceBuffer(depth.nbytes, dev).upload(depth); bufs.append(d_depth)
            d_kern = DeviceBuffer(self._uh.kernel.nbytes, dev).upload(self._uh.kernel); bufs.append(d_kern)
            d_state = DeviceBuffer(self._uh.kernel.nbytes, dev).upload(np.ascontiguousarray(self._uh.state)); bufs.append(d_state)
            d_conv = DeviceBuffer(depth.nbytes, dev); bufs.append(d_conv)
            uh_convolve_dev(d_kern, d_state, d_depth, d_conv, T, n_ks, n, device=dev)
            self._uh.state = d_state.download(np.float64, self._uh.kernel.shape)
            d_depth.free()
            q_ch = np.array(self.channel_state[self.inner_idx], dtype=np.float64, order='C')
            d_qch = DeviceBuffer(max(q_ch.nbytes, 8), dev).upload(q_ch); bufs.append(d_qch)
            d_qfull = DeviceBuffer(max(q_ch.nbytes, 8), dev).upload(q_ch); bufs.append(d_qfull)
            d_out = DeviceBuffer(depth.nbytes, dev); bufs.append(d_out)
            d_f32 = DeviceBuffer((T // rows_per_output) * n * 4, dev); bufs.append(d_f32)
            self._plan.unit_route_dev(d_qch, d_qfull, d_conv, T, d_out, T, T, self.num_routing_steps_per_runoff)
            resample_cast_dev(d_out, T, n, rows_per_output, d_f32, dev)
            q_array = d_f32.download(np.float32, (T // rows_per_output, n))
            q_final = d_conv.download(np.float64, (n,), offset=(T - 1) * n * 8)    # headwaters keep the last lateral row
            q_final[self.inner_idx] = d_qfull.download(np.float64, q_ch.shape)
        finally:
            for b in bufs:
                b.free()
        return q_final, q_array

    def _write_final_state(self) -> None:
        super()._write_final_state()
        if self.cfg.uh_state_final_file and self._uh is not None:
            self._uh.write_state(self.cfg.uh_state_final_file)

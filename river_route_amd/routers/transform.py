"""
`TransformMuskingum`: the shared per-file driver of the routers that take lateral inflow
(river_route/routers/TransformMuskingum.py:14-152): input generator, time-step algebra, sequential / ensemble
state hand-off, resampling to dt_discharge and the float32 cast before the writer.
"""
from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np

from .muskingum import Muskingum

__all__ = ['TransformMuskingum']


class TransformMuskingum(Muskingum, ABC):
    _ROUTER_REQUIRED_CONFIGS: tuple[str, ...] = ()
    _as_volumes: bool = False
    _device_postprocess: bool = True   # keep each file on the GPU from lateral upload to float32 discharge

    def _qlateral_generator(self):
        """Yields (dates datetime64[s], lateral float64 (T, n), input file, output file) per input file."""
        if self.cfg.qlateral_files:
            from ..io import read_qlateral
            for lateral_file, discharge_file in zip(self.cfg.qlateral_files, self.cfg.discharge_files):
                self.logger.info('-' * 60)
                dates, array = read_qlateral(lateral_file, self.cfg.var_t)
                yield dates, array, lateral_file, discharge_file
        elif self.cfg.grid_runoff_files and self.cfg.grid_weights_file:
            # gridded runoff -> catchment inflow on the device (TransformMuskingum.py:38-51 -> runoff.runoff_to_qlateral)
            from ..runoff import runoff_to_qlateral
            for runoff_file, discharge_file in zip(self.cfg.grid_runoff_files, self.cfg.discharge_files):
                self.logger.info('-' * 60)
                self.logger.debug(f'Calculating qlateral: {runoff_file}')
                ds = runoff_to_qlateral(runoff_file, grid_weights_file=self.cfg.grid_weights_file,
                                        var_runoff=self.cfg.var_grid_runoff, var_x=self.cfg.var_x, var_y=self.cfg.var_y,
                                        var_t=self.cfg.var_t, var_river_id=self.cfg.var_river_id,
                                        cumulative=self.cfg.grid_accumulation_type == 'cumulative',
                                        as_volumes=self._as_volumes, device=self.cfg.device)
                yield (ds['time'].values.astype('datetime64[s]'),
                       ds['qlateral'].values.astype(np.float64, copy=False), runoff_file, discharge_file)

    def _validate_router_configs(self) -> None:
        qlateral = self.cfg.qlateral_files
        grids = self.cfg.grid_runoff_files and self.cfg.grid_weights_file
        if qlateral and grids:
            raise ValueError('Provide qlateral_files or grid_runoff_files with grid_weights_file, not both')
        if not qlateral and not grids:
            raise ValueError('Provide qlateral_files or grid_runoff_files with grid_weights_file')
        n_inputs = len(qlateral) + len(self.cfg.grid_runoff_files or [])
        if len(self.cfg.discharge_files) != n_inputs:
            raise ValueError('Number of resolved discharge output files must match number of input files')

    def _set_network_and_time_dependent_vectors(self, dates: np.ndarray) -> None:
        """Time-step defaults and rules of docs/references/time-options.md (TransformMuskingum.py:66-106)."""
        self.logger.debug('Setting and validating time parameters')
        self.dt_runoff = self.cfg.dt_runoff or (dates[1] - dates[0]).astype('timedelta64[s]').astype(int)
        self.dt_discharge = self.cfg.dt_discharge or self.dt_runoff
        self.dt_total = self.cfg.dt_total or self.dt_runoff * dates.shape[0]
        if not self.cfg.dt_routing:
            self.logger.warning('dt_routing was not provided or is Null/False, defaulting to dt_runoff')
        self.dt_routing = self.cfg.dt_routing or self.dt_runoff

        signature = (self.dt_total, self.dt_runoff, self.dt_discharge, self.dt_routing)
        if self._network_time_signature == signature:
            return
        for big, small in (('dt_total', 'dt_runoff'), ('dt_total', 'dt_discharge'), ('dt_discharge', 'dt_runoff'),
                           ('dt_runoff', 'dt_routing')):
            if getattr(self, big) < getattr(self, small):
                raise ValueError(f'{big} must be >= {small}')
        for big, small in (('dt_total', 'dt_runoff'), ('dt_total', 'dt_discharge'), ('dt_discharge', 'dt_runoff'),
                           ('dt_runoff', 'dt_routing')):
            if getattr(self, big) % getattr(self, small) != 0:
                raise ValueError(f'{big} must be an integer multiple of {small}')
        self.num_runoff_steps = int(self.dt_total / self.dt_runoff)
        self.num_runoff_steps_per_discharge = int(self.dt_discharge / self.dt_runoff)
        self.num_routing_steps_per_runoff = int(self.dt_runoff / self.dt_routing)
        self._set_muskingum_coefficients(self.dt_routing)
        self.c4 = self.c1 + self.c2
        self._network_time_signature = signature

    def _execute_routing(self) -> None:
        self._ensemble_member_states = []
        total_files = len(self.cfg.qlateral_files or self.cfg.grid_runoff_files)
        file_iter = self._qlateral_generator()
        if self.cfg.progress_bar:
            from tqdm import tqdm
            file_iter = tqdm(file_iter, total=total_files, desc='Files Routed')

        for dates, qlateral, runoff_file, discharge_file in file_iter:
            self.logger.info(f'Routing qlateral: {runoff_file}')
            self._set_network_and_time_dependent_vectors(dates)
            self.logger.debug('Starting routing computation')
            per = int(self.dt_discharge / self.dt_runoff) if self.dt_discharge > self.dt_runoff else 1
            if self._device_postprocess and hasattr(self._plan, 'rapid_route_dev'):
                # device-resident file: lateral in once, resample-mean + float32 cast on the GPU, float32 out
                q_t, q_array = self._router_device(qlateral, per)
            else:
                q_t, q_array = self._router(qlateral)
                if per > 1:
                    q_array = q_array.reshape((int(self.dt_total / self.dt_discharge), per, self.A.shape[0])).mean(axis=1)
                q_array = q_array.astype(np.float32, copy=False)
            if self.cfg.runoff_processing_mode == 'sequential':
                self.channel_state = q_t
            elif self.cfg.runoff_processing_mode == 'ensemble':
                self._ensemble_member_states.append(q_t.copy())
            if per > 1:
                self.logger.debug('Resampling dates and discharges to specified timestep')
                dates = dates[::self.num_runoff_steps_per_discharge]

            self.logger.debug('Writing Discharge Array to File')
            self._write_discharges(dates, q_array, discharge_file, runoff_file)

        if self.cfg.runoff_processing_mode == 'ensemble':
            self.channel_state = np.array(self._ensemble_member_states).mean(axis=0)
        self.logger.info('-' * 60)

    @abstractmethod
    def _router(self, qlateral: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        """(final state float64[n], discharge float64[T, n]) for one input file, host arrays."""

    @abstractmethod
    def _router_device(self, qlateral: np.ndarray, rows_per_output: int) -> tuple[np.ndarray, np.ndarray]:
        """(final state float64[n], discharge float32[T / rows_per_output, n]) with the whole file on the device."""

    def _check_lateral(self, qlateral: np.ndarray) -> np.ndarray:
        ql = np.ascontiguousarray(qlateral, dtype=np.float64)
        if ql.shape != (self.num_runoff_steps, self.A.shape[0]):
            raise ValueError(f'lateral inflow has shape {ql.shape}, expected '
                             f'({self.num_runoff_steps}, {self.A.shape[0]}) from the time options')
        return ql

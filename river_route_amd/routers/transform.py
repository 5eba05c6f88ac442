"""
`TransformMuskingum`: the shared per-file driver of the routers that take lateral inflow
(river_route/routers/TransformMuskingum.py:14-152): input generator, time-step algebra, sequential / ensemble
state hand-off, resampling to dt_discharge and the float32 cast before the writer.
"""
from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np

from .muskingum import PROGRESS, Muskingum

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
                rows = self._file_rows(lateral_file)
                if rows is not None:      # a NetCDF-3 float32 file: its rows go from the page cache to the device as they are (nc3.py)
                    yield rows[0], rows[1], lateral_file, discharge_file
                    continue
                # a router that converts on the device takes a float32 file as float32 (RapidMuskingum: rr_rapid_route_f32in_dev)
                keep32 = self._device_postprocess and hasattr(self, '_route_on_device_f32in') and \
                    type(self)._router is getattr(type(self), '_engine_router', None)
                dates, array = read_qlateral(lateral_file, self.cfg.var_t, keep_float32=keep32)
                yield dates, array, lateral_file, discharge_file
        elif self.cfg.grid_runoff_files and self.cfg.grid_weights_file:
            # gridded runoff -> catchment inflow on the device (TransformMuskingum.py:38-51 -> runoff.runoff_to_qlateral).  A
            # router that can take the runoff itself (`_router_device_runoff`) gets the prepared source instead of the
            # (time, river) array, so the inflow is computed on its way into the engine's records
            from ..runoff import prepare_runoff, runoff_to_qlateral
            kw = dict(grid_weights_file=self.cfg.grid_weights_file, var_runoff=self.cfg.var_grid_runoff, var_x=self.cfg.var_x,
                      var_y=self.cfg.var_y, var_t=self.cfg.var_t, var_river_id=self.cfg.var_river_id,
                      cumulative=self.cfg.grid_accumulation_type == 'cumulative', device=self.cfg.device)
            for runoff_file, discharge_file in zip(self.cfg.grid_runoff_files, self.cfg.discharge_files):
                self.logger.info('-' * 60)
                self.logger.debug(f'Calculating qlateral: {runoff_file}')
                if self._takes_runoff_source():
                    src = prepare_runoff(runoff_file, **kw)
                    if not src.irregular:
                        yield src.time_index.astype('datetime64[s]'), src, runoff_file, discharge_file
                        continue
                ds = runoff_to_qlateral(runoff_file, as_volumes=self._as_volumes, **kw)
                yield (ds['time'].values.astype('datetime64[s]'),
                       ds['qlateral'].values.astype(np.float64, copy=False), runoff_file, discharge_file)

    def _file_rows(self, lateral_file):
        """(dates, nc3.RowBlock) where the router can take a qlateral file's rows straight from the file -- a flat (NetCDF-3) float32
        variable, the router's own engine path and the default discharge writer -- else None."""
        if not (self._device_postprocess and hasattr(self, '_route_file_to_file') and hasattr(self._plan, 'rapid_route_dev')
                and type(self)._router is getattr(type(self), '_engine_router', None) and '_write_discharges' not in self.__dict__
                and type(self)._write_discharges is Muskingum._write_discharges):
            return None
        from .. import nc3
        from ..io import _decode_cf_time
        blk = nc3.locate_rows(lateral_file, 'qlateral')
        if blk is None or blk.dtype.kind != 'f' or blk.dtype.itemsize != 4 or blk.cols != self.A.shape[0] or blk.rows < 2:
            return None
        try:
            values, atts = nc3.read_vector(lateral_file, self.cfg.var_t)
            return _decode_cf_time(values, atts['units']), blk
        except (KeyError, ValueError):
            return None

    def _takes_runoff_source(self) -> bool:
        return (self._device_postprocess and hasattr(self, '_route_on_device') and hasattr(self._plan, 'rapid_route_dev')
                and type(self)._router is getattr(type(self), '_engine_router', None))

    def _router_device_runoff(self, source, rows_per_output: int) -> tuple[np.ndarray, np.ndarray]:
        """Gridded runoff of one file routed without leaving the GPU: the runoff block and the weight table go up once,
        rr_runoff_to_qlateral_dev writes the catchment inflow as device rows (volumes for RapidMuskingum, depths for
        UnitMuskingum), and the router's device path takes it from there.  RapidMuskingum overrides this with the call that
        computes the inflow inside the record pass (rr_rapid_route_runoff_dev) where that applies."""
        from ..engine import runoff_to_qlateral_dev
        from ._device import Arena
        T, n = source.runoff_tp.shape[0], self.A.shape[0]
        if source.river_ids.shape[0] != n or T != self.num_runoff_steps:
            raise ValueError(f'gridded runoff covers {source.river_ids.shape[0]} rivers x {T} steps, the network and time options '
                             f'expect {n} x {self.num_runoff_steps}')
        block = source.point_major()
        with Arena(self.cfg.device) as arena:
            d_block, d_ptr, d_idx = arena.put(block), arena.put(source.indptr), arena.put(source.indices)
            d_w = arena.put(source.weights)
            d_area = arena.put(source.area) if self._as_volumes else None
            d_lat = arena.empty(T * n * 8)
            runoff_to_qlateral_dev(n, block.shape[0], T, d_ptr, d_idx, d_w, d_block, block.dtype == np.float32, 1, block.shape[1], d_area,
                                   source.flags, d_lat, device=self.cfg.device)
            arena.release(d_block)
            return self._route_on_device(arena, d_lat, T, rows_per_output)

    def _validate_router_configs(self) -> None:
        laterals = list(self.cfg.qlateral_files or [])
        grids = list(self.cfg.grid_runoff_files or []) if self.cfg.grid_weights_file else []
        if laterals and self.cfg.grid_runoff_files and self.cfg.grid_weights_file:
            raise ValueError('Provide qlateral_files or grid_runoff_files with grid_weights_file, not both')
        if not laterals and not grids:
            raise ValueError('Provide qlateral_files or grid_runoff_files with grid_weights_file')
        if len(self.cfg.discharge_files) != len(laterals) + len(self.cfg.grid_runoff_files or []):
            raise ValueError('Number of resolved discharge output files must match number of input files')

    # the four time steps, coarsest first, and the rule each adjacent pair obeys (docs/references/time-options.md:32-50)
    _STEP_PAIRS = (('dt_total', 'dt_runoff'), ('dt_total', 'dt_discharge'), ('dt_discharge', 'dt_runoff'), ('dt_runoff', 'dt_routing'))

    def _set_network_and_time_dependent_vectors(self, dates: np.ndarray) -> None:
        """Time-step defaults and rules (river_route/routers/TransformMuskingum.py:66-106), then the coefficients."""
        self.logger.debug('Setting and validating time parameters')
        cfg = self.cfg
        from_dates = int((dates[1] - dates[0]).astype('timedelta64[s]').astype(int)) if not cfg.dt_runoff else None
        self.dt_runoff = cfg.dt_runoff or from_dates
        self.dt_discharge = cfg.dt_discharge or self.dt_runoff
        self.dt_total = cfg.dt_total or self.dt_runoff * dates.shape[0]
        if not cfg.dt_routing:
            self.logger.warning('dt_routing was not provided or is Null/False, defaulting to dt_runoff')
        self.dt_routing = cfg.dt_routing or self.dt_runoff

        steps = (self.dt_total, self.dt_runoff, self.dt_discharge, self.dt_routing)
        if steps == self._network_time_signature:
            return      # same grid as the previous file: counts and coefficients stand
        value = dict(zip(('dt_total', 'dt_runoff', 'dt_discharge', 'dt_routing'), steps))
        for rule, message in ((lambda a, b: a >= b, '{} must be >= {}'), (lambda a, b: a % b == 0, '{} must be an integer multiple of {}')):
            for coarse, fine in self._STEP_PAIRS:
                if not rule(value[coarse], value[fine]):
                    raise ValueError(message.format(coarse, fine))
        self.num_runoff_steps = self.dt_total // self.dt_runoff
        self.num_runoff_steps_per_discharge = self.dt_discharge // self.dt_runoff
        self.num_routing_steps_per_runoff = self.dt_runoff // self.dt_routing
        self._set_muskingum_coefficients(self.dt_routing)
        self.c4 = self.c1 + self.c2
        self._network_time_signature = steps

    def _route_one_file(self, qlateral: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        """(final state, float32 discharge at dt_discharge) of one file.  The file stays on the GPU from the lateral
        upload to the float32 rows unless a subclass brings its own `_router` (the reference's extension point), the
        engine is not the HIP plan, or the file does not fit on the card: then `_router` + the post-processing of
        TransformMuskingum.py:128-142 on the host."""
        per = self.num_runoff_steps_per_discharge
        from ..runoff import RunoffSource
        if isinstance(qlateral, RunoffSource):
            from .._lib import RR_E_UNSUPPORTED, RRError
            from ._device import DeviceOutOfMemory
            try:
                return self._router_device_runoff(qlateral, per)
            except (DeviceOutOfMemory, RRError) as e:
                if isinstance(e, RRError) and e.code != RR_E_UNSUPPORTED:
                    raise
                qlateral = qlateral.to_array(self._as_volumes)      # the two-step form: (time, river) rows, then the routing call
        own_router = type(self)._router is getattr(type(self), '_engine_router', None)
        if self._device_postprocess and own_router and hasattr(self._plan, 'rapid_route_dev'):
            from ._device import DeviceOutOfMemory
            try:
                return self._router_device(qlateral, per)
            except DeviceOutOfMemory as e:
                self.logger.warning(f'file does not fit on the device ({e}); routing it in chunks from host memory')
        q_t, q_array = self._router(qlateral)
        if per > 1:
            q_array = q_array.reshape((-1, per, q_array.shape[1])).mean(axis=1)
        return q_t, q_array.astype(np.float32, copy=False)

    def _execute_routing(self) -> None:
        import time
        members: list[np.ndarray] = []
        files = self._qlateral_generator()
        if self.cfg.progress_bar:
            from tqdm import tqdm
            files = tqdm(files, total=len(self.cfg.qlateral_files or self.cfg.grid_runoff_files), desc='Files Routed')
        sequential = self.cfg.runoff_processing_mode == 'sequential'
        for dates, qlateral, runoff_file, discharge_file in files:
            self.logger.info(f'Routing qlateral: {runoff_file}')
            self._set_network_and_time_dependent_vectors(dates)
            self.logger.debug('Starting routing computation')
            t0 = time.perf_counter()
            from ..nc3 import RowBlock
            written = False
            if isinstance(qlateral, RowBlock):
                q_t = self._route_file_to_file(qlateral, dates[::self.num_runoff_steps_per_discharge], discharge_file, runoff_file)
                written = q_t is not None
                if not written:      # the fused form does not apply to this call, or the file does not fit the card: the file as an array
                    from ..io import read_qlateral
                    qlateral = read_qlateral(runoff_file, self.cfg.var_t, keep_float32=True)[1]
            if not written:
                q_t, q_array = self._route_one_file(qlateral)
            seconds = time.perf_counter() - t0
            reach_steps = self.A.shape[0] * self.num_runoff_steps * self.num_routing_steps_per_runoff
            self.logger.log(PROGRESS, f'{reach_steps / max(seconds, 1e-9):.3e} reach-steps/s '
                                      f'({reach_steps * 16 / max(seconds, 1e-9) / 1e9:.1f} GB/s of lateral + discharge rows) for {runoff_file}')
            if sequential:
                self.channel_state = q_t
            else:
                members.append(np.array(q_t, copy=True))
            if self.num_runoff_steps_per_discharge > 1:
                self.logger.debug('Resampling dates and discharges to specified timestep')
                dates = dates[::self.num_runoff_steps_per_discharge]
            if not written:
                self.logger.debug('Writing Discharge Array to File')
                self._write_discharges(dates, q_array, discharge_file, runoff_file)
        if not sequential:
            self._ensemble_member_states = members
            self.channel_state = np.mean(np.array(members), axis=0)
        self.logger.info('-' * 60)

    @abstractmethod
    def _router(self, qlateral: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        """(final state float64[n], discharge float64[T, n]) for one input file, host arrays."""

    @abstractmethod
    def _router_device(self, qlateral: np.ndarray, rows_per_output: int) -> tuple[np.ndarray, np.ndarray]:
        """(final state float64[n], discharge float32[T / rows_per_output, n]) with the whole file on the device."""

    def _check_lateral(self, qlateral: np.ndarray, keep_float32: bool = False) -> np.ndarray:
        qlateral = np.asarray(qlateral)
        ql = np.ascontiguousarray(qlateral, dtype=np.float32 if keep_float32 and qlateral.dtype == np.float32 else np.float64)
        if ql.shape != (self.num_runoff_steps, self.A.shape[0]):
            raise ValueError(f'lateral inflow has shape {ql.shape}, expected '
                             f'({self.num_runoff_steps}, {self.A.shape[0]}) from the time options')
        return ql

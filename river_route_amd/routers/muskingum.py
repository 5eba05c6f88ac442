"""
`Muskingum`: channel-only Muskingum routing behind the reference's Router API
(river_route/routers/Muskingum.py:23-352): same constructor, config keys, attributes, hooks, exceptions and
`.route()` lifecycle; the routing loop itself runs on the GPU through librr_hip.so (no CPU fallback).
"""
from __future__ import annotations

import datetime
import json
import logging
import sys
import traceback
from typing import Any

import numpy as np

from ..engine import Plan
from ..tools import adjacency_matrix
from .config import Configs

__all__ = ['Muskingum', 'PROGRESS']

PROGRESS = 25   # custom log level of the reference (river_route/logging.py:3-4)
logging.addLevelName(PROGRESS, 'PROGRESS')


class Muskingum:
    cfg: Configs
    logger: logging.Logger

    # keys a router needs non-null, checked at route() time (Muskingum.py:36, 98-104)
    _ROUTER_REQUIRED_CONFIGS: tuple[str, ...] = ('channel_state_init_file', 'dt_routing', 'dt_total')
    _network_time_signature: tuple | None = None

    def __init__(self, configs: Any = None, **kwargs: Any) -> None:
        raw: dict[str, Any] = {}
        if configs is not None and configs != '':
            if isinstance(configs, Configs):
                raw = configs.as_dict()
            elif str(configs).endswith('.json'):
                with open(configs, 'r') as f:
                    raw = json.load(f)
            elif str(configs).endswith(('.yml', '.yaml')):
                import yaml
                with open(configs, 'r') as f:
                    raw = yaml.load(f, Loader=yaml.FullLoader)
            else:
                raise RuntimeError('Unrecognized simulation config file type. Must be .json or .yaml')
        raw.update(kwargs)
        raw.pop('_router', None)
        self.cfg = Configs(**raw)

        self.logger = logging.getLogger(f'river_route.{id(self):x}')
        self.logger.disabled = not self.cfg.log
        self.logger.setLevel(self.cfg.log_level)
        handler: logging.Handler
        if self.cfg.log_stream == 'stdout':
            handler = logging.StreamHandler(sys.stdout)
        else:
            handler = logging.FileHandler(self.cfg.log_stream)
        handler.setFormatter(logging.Formatter(self.cfg.log_format))
        self.logger.addHandler(handler)
        self.logger.debug('Logger initialized')
        self._plan: Plan | None = None
        self._coeffs_on_device: tuple | None = None

    def __repr__(self) -> str:
        return f'{type(self).__name__}(params_file={self.cfg.params_file!r})'

    # ------------------------------------------------------------------ validation
    def _validate_configs(self) -> None:
        self.logger.debug('Validating configs file')
        for key in self._ROUTER_REQUIRED_CONFIGS:
            if not getattr(self.cfg, key, None):
                raise ValueError(f'{key} is required for {type(self).__name__}')
        self._validate_router_configs()

    def _validate_router_configs(self) -> None:
        if len(self.cfg.discharge_files) != 1:
            raise ValueError('Muskingum requires exactly one entry in discharge_files')

    # ------------------------------------------------------------------ state
    def _read_initial_state(self) -> None:
        if hasattr(self, 'channel_state'):
            return   # a second route() continues from the state the first one left (Muskingum.py:117-118)
        state_file = self.cfg.channel_state_init_file
        if not state_file:
            self.logger.warning('channel_state_init_file not provided. Defaulting to zero initial conditions')
            self.channel_state = np.zeros(self.A.shape[0], dtype=np.float64)
            return
        import pandas as pd
        self.logger.debug('Reading Initial State from Parquet')
        self.channel_state = pd.read_parquet(state_file).values.flatten().astype(np.float64, copy=False)

    def _write_final_state(self) -> None:
        if not self.cfg.channel_state_final_file:
            return
        import pandas as pd
        self.logger.debug('Writing Final State to Parquet')
        pd.DataFrame({'Q': self.channel_state}).to_parquet(self.cfg.channel_state_final_file)

    # ------------------------------------------------------------------ network + coefficients
    def _set_network_dependent_vectors(self) -> None:
        import pandas as pd
        self.logger.debug('Calculating network dependent vectors')
        try:
            df = pd.read_parquet(self.cfg.params_file,
                                 columns=[self.cfg.var_river_id, 'k', 'x', 'downstream_river_id'])
        except Exception as e:
            self.logger.error(f'Error reading required parameter columns from params_file: {e}')
            self.logger.debug(traceback.format_exc())
            raise
        if df[self.cfg.var_river_id].duplicated().any():
            raise ValueError('params_file contains duplicate river IDs.')
        self.river_ids = df[self.cfg.var_river_id].to_numpy(dtype=np.int64, copy=False)
        downstream = df['downstream_river_id'].to_numpy(dtype=np.int64, copy=False)
        self.k = df['k'].to_numpy(dtype=np.float64, copy=False)
        self.x = df['x'].to_numpy(dtype=np.float64, copy=False)
        unknown = np.setdiff1d(downstream[downstream > 0], self.river_ids)
        if unknown.size:
            raise ValueError(f'params_file has downstream IDs not in river_id column: {unknown[:10].tolist()}')
        self.A = adjacency_matrix(self.river_ids, downstream)
        if self._plan is not None:
            self._plan.close()
        self._plan = Plan(self.A.indptr, self.A.indices, device=self.cfg.device)
        self._coeffs_on_device = None
        self.logger.log(PROGRESS, f'Network: {self.A.shape[0]} river segments')

    def _set_muskingum_coefficients(self, dt_routing: float) -> None:
        """c1, c2, c3 from k, x and the routing step (river_route/routers/Muskingum.py:172-193)."""
        self.logger.debug('Calculating Muskingum coefficients')
        with np.errstate(divide='ignore', invalid='ignore'):
            ratio = dt_routing / self.k
            twice_x = 2 * self.x
            denom = ratio + (2 * (1 - self.x))
            self.c1 = (ratio - twice_x) / denom
            self.c2 = (ratio + twice_x) / denom
            self.c3 = ((2 * (1 - self.x)) - ratio) / denom
        if not np.allclose(self.c1 + self.c2 + self.c3, 1):
            self.logger.warning('Muskingum coefficients do not sum to 1')
            raise ValueError('Muskingum coefficients do not sum to 1, check routing parameters and time step')
        csc = self.A.tocsc()
        self._csc_indptr = csc.indptr
        self._csc_indices = csc.indices
        self._lhs_off_data = np.ascontiguousarray(-self.c1[csc.indices])
        self._coeffs_on_device = None

    def _upload_coefficients(self, c4_dt: np.ndarray | None, tag: tuple) -> None:
        """Coefficients stay resident on the GPU until they are recomputed or `tag` (router kind, dt_runoff) changes."""
        if self._coeffs_on_device == tag:
            return
        self._plan.set_coeffs(self._lhs_off_data, self.c2, self.c3, c4_dt)
        self._coeffs_on_device = tag

    # ------------------------------------------------------------------ lifecycle
    def route(self):
        """Run the simulation described by the configs; returns self with `channel_state` updated and the
        discharge handed to the writer."""
        self.logger.log(PROGRESS, 'Beginning routing')
        t1 = datetime.datetime.now()
        self._validate_configs()
        self.logger.debug(self)
        self._set_network_dependent_vectors()
        self._read_initial_state()
        self._hook_before_route()
        self._execute_routing()
        self._write_final_state()
        self._hook_after_route()
        t2 = datetime.datetime.now()
        self.logger.log(PROGRESS, f'Routing completed in {(t2 - t1).total_seconds()} seconds')
        return self

    def _execute_routing(self) -> None:
        import pandas as pd
        self.logger.info('-' * 60)
        self.dt_routing = self.cfg.dt_routing
        self.dt_total = self.cfg.dt_total
        self.dt_discharge = self.cfg.dt_discharge or self.dt_routing
        if not (self.dt_total >= self.dt_discharge >= self.dt_routing):
            raise ValueError('Need dt_total >= dt_discharge >= dt_routing')
        if self.dt_total % self.dt_discharge != 0:
            raise ValueError('dt_total must be an integer multiple of dt_discharge')
        if self.dt_discharge % self.dt_routing != 0:
            raise ValueError('dt_discharge must be an integer multiple of dt_routing')
        num_output_steps = int(self.dt_total / self.dt_discharge)
        num_routing_per_output = int(self.dt_discharge / self.dt_routing)
        self._set_muskingum_coefficients(self.dt_routing)

        self.logger.debug('Starting routing computation')
        discharge_array = self._router(num_output_steps, num_routing_per_output)
        dates = pd.date_range(start=self.cfg.start_datetime, periods=num_output_steps,
                              freq=pd.to_timedelta(self.dt_discharge, unit='s')).to_numpy()
        self.logger.debug('Writing Discharge Array to File')
        discharge_array = discharge_array.astype(np.float32, copy=False)
        self._write_discharges(dates, discharge_array, self.cfg.discharge_files[0])
        self.logger.info('-' * 60)

    def _router(self, num_output_steps: int, num_routing_per_output: int) -> np.ndarray:
        """(I - c1 A) Q(t+1) = c2 (A Q(t)) + c3 Q(t), no lateral inflow; rr_muskingum_route."""
        q_init = self.channel_state
        if not np.any(q_init):
            self.logger.warning(
                'Initial channel state is all zeros. Muskingum routing without lateral inflow requires a '
                'non-zero initial state to produce meaningful results. Provide channel_state_init_file.')
        n = self.A.shape[0]
        discharge_array = np.zeros((num_output_steps, n), dtype=np.float64)
        q_t = np.array(q_init, dtype=np.float64, order='C')
        self._upload_coefficients(None, ('muskingum',))
        self._plan.muskingum_route(q_t, discharge_array, num_output_steps, num_routing_per_output)
        self.channel_state = q_t
        return discharge_array

    # ------------------------------------------------------------------ hooks + dependency injection
    def _hook_before_route(self) -> None:
        return

    def _hook_after_route(self) -> None:
        return

    def set_write_discharges(self, func):
        """Replace the discharge writer: func(dates, q_array, q_file, routed_file='') (types.py:17-24)."""
        self._write_discharges = func
        return self

    def _write_discharges(self, dates, q_array, q_file, routed_file='') -> None:
        from ..io import write_discharge
        write_discharge(q_file, dates, q_array, self.river_ids, self.cfg.var_river_id, self.cfg.var_discharge,
                        routed_file)

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


def _load_config_source(configs: Any, overrides: dict[str, Any]) -> Configs:
    """Configs from a file path (.json / .yml / .yaml), a Configs object or nothing, with keyword overrides on top
    (river_route/routers/Muskingum.py:67-81)."""
    if isinstance(configs, Configs):
        if not overrides:
            return configs                      # already validated; paths already resolved
        raw = configs.as_dict()
        if raw.get('discharge_dir'):
            raw.pop('discharge_files', None)     # derived from discharge_dir by the first validation: derive it again
    elif configs is None or configs == '':
        raw = {}
    else:
        name = str(configs)
        if name.endswith('.json'):
            with open(configs, 'r') as f:
                raw = json.load(f)
        elif name.endswith(('.yml', '.yaml')):
            import yaml
            with open(configs, 'r') as f:
                raw = yaml.load(f, Loader=yaml.FullLoader)
        else:
            raise RuntimeError('Unrecognized simulation config file type. Must be .json or .yaml')
    raw = {**raw, **overrides}
    raw.pop('_router', None)
    return Configs(**raw)


class Muskingum:
    cfg: Configs
    logger: logging.Logger

    # keys a router needs non-null, checked at route() time (Muskingum.py:36, 98-104)
    _ROUTER_REQUIRED_CONFIGS: tuple[str, ...] = ('channel_state_init_file', 'dt_routing', 'dt_total')
    _network_time_signature: tuple | None = None

    def __init__(self, configs: Any = None, **kwargs: Any) -> None:
        self.cfg = _load_config_source(configs, kwargs)
        self.logger = self._make_logger()
        self.logger.debug('Logger initialized')
        self._plan: Plan | None = None
        self._coeffs_on_device: tuple | None = None

    def _make_logger(self) -> logging.Logger:
        """One logger per router object, `river_route.<id>`, at the configured level and stream (Muskingum.py:84-92)."""
        log = logging.getLogger(f'river_route.{id(self):x}')
        log.disabled = not self.cfg.log
        log.setLevel(self.cfg.log_level)
        sink = logging.StreamHandler(sys.stdout) if self.cfg.log_stream == 'stdout' else logging.FileHandler(self.cfg.log_stream)
        sink.setFormatter(logging.Formatter(self.cfg.log_format))
        log.addHandler(sink)
        return log

    def __repr__(self) -> str:
        return f'{type(self).__name__}(params_file={self.cfg.params_file!r})'

    # ------------------------------------------------------------------ validation
    def _validate_configs(self) -> None:
        self.logger.debug('Validating configs file')
        missing = [key for key in self._ROUTER_REQUIRED_CONFIGS if not getattr(self.cfg, key, None)]
        if missing:
            raise ValueError(f'{missing[0]} is required for {type(self).__name__}')
        self._validate_router_configs()

    def _validate_router_configs(self) -> None:
        if len(self.cfg.discharge_files) != 1:
            raise ValueError('Muskingum requires exactly one entry in discharge_files')

    # ------------------------------------------------------------------ state
    def _read_initial_state(self) -> None:
        if hasattr(self, 'channel_state'):
            return   # a second route() continues from the state the first one left (Muskingum.py:117-118)
        n = self.A.shape[0]
        if self.cfg.channel_state_init_file:
            import pandas as pd
            self.logger.debug('Reading Initial State from Parquet')
            self.channel_state = pd.read_parquet(self.cfg.channel_state_init_file).to_numpy(dtype=np.float64).reshape(-1)
        else:
            self.logger.warning('channel_state_init_file not provided. Defaulting to zero initial conditions')
            self.channel_state = np.zeros(n, dtype=np.float64)

    def _write_final_state(self) -> None:
        target = self.cfg.channel_state_final_file
        if target:
            import pandas as pd
            self.logger.debug('Writing Final State to Parquet')
            pd.DataFrame({'Q': self.channel_state}).to_parquet(target)

    # ------------------------------------------------------------------ network + coefficients
    def _set_network_dependent_vectors(self) -> None:
        """ids, k, x and the adjacency from the params file (Muskingum.py:141-169); the engine's plan of the network."""
        import pandas as pd
        self.logger.debug('Calculating network dependent vectors')
        id_col = self.cfg.var_river_id
        try:
            table = pd.read_parquet(self.cfg.params_file, columns=[id_col, 'k', 'x', 'downstream_river_id'])
        except Exception as e:
            self.logger.error(f'Error reading required parameter columns from params_file: {e}')
            self.logger.debug(traceback.format_exc())
            raise
        ids = table[id_col].to_numpy(dtype=np.int64, copy=False)
        if np.unique(ids).size != ids.size:
            raise ValueError('params_file contains duplicate river IDs.')
        downstream = table['downstream_river_id'].to_numpy(dtype=np.int64, copy=False)
        strays = np.setdiff1d(downstream[downstream > 0], ids)
        if strays.size:
            raise ValueError(f'params_file has downstream IDs not in river_id column: {strays[:10].tolist()}')
        self.river_ids = ids
        self.k = table['k'].to_numpy(dtype=np.float64, copy=False)
        self.x = table['x'].to_numpy(dtype=np.float64, copy=False)
        self.A = adjacency_matrix(ids, downstream)
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
        started = datetime.datetime.now()
        self.logger.log(PROGRESS, 'Beginning routing')
        self._validate_configs()
        self.logger.debug(self)
        for stage in (self._set_network_dependent_vectors, self._read_initial_state, self._hook_before_route,
                      self._execute_routing, self._write_final_state, self._hook_after_route):
            stage()
        self.logger.log(PROGRESS, f'Routing completed in {(datetime.datetime.now() - started).total_seconds()} seconds')
        return self

    def _execute_routing(self) -> None:
        """Channel-only run: steps from the configs, one routing call, float32 rows to the writer (Muskingum.py:229-260)."""
        import pandas as pd
        self.logger.info('-' * 60)
        self.dt_routing, self.dt_total = self.cfg.dt_routing, self.cfg.dt_total
        self.dt_discharge = self.cfg.dt_discharge or self.dt_routing
        if not (self.dt_total >= self.dt_discharge >= self.dt_routing):
            raise ValueError('Need dt_total >= dt_discharge >= dt_routing')
        for coarse, fine in (('dt_total', 'dt_discharge'), ('dt_discharge', 'dt_routing')):
            if getattr(self, coarse) % getattr(self, fine) != 0:
                raise ValueError(f'{coarse} must be an integer multiple of {fine}')
        rows, per_row = self.dt_total // self.dt_discharge, self.dt_discharge // self.dt_routing
        self._set_muskingum_coefficients(self.dt_routing)
        self.logger.debug('Starting routing computation')
        routed = self._router(int(rows), int(per_row))
        stamps = pd.date_range(start=self.cfg.start_datetime, periods=int(rows),
                               freq=pd.to_timedelta(self.dt_discharge, unit='s')).to_numpy()
        self.logger.debug('Writing Discharge Array to File')
        self._write_discharges(stamps, routed.astype(np.float32, copy=False), self.cfg.discharge_files[0])
        self.logger.info('-' * 60)

    def _router(self, num_output_steps: int, num_routing_per_output: int) -> np.ndarray:
        """(I - c1 A) Q(t+1) = c2 (A Q(t)) + c3 Q(t), no lateral inflow; rr_muskingum_route."""
        state = np.array(self.channel_state, dtype=np.float64, order='C')
        if not state.any():
            self.logger.warning(
                'Initial channel state is all zeros. Muskingum routing without lateral inflow requires a '
                'non-zero initial state to produce meaningful results. Provide channel_state_init_file.')
        routed = np.zeros((num_output_steps, state.shape[0]), dtype=np.float64)
        self._upload_coefficients(None, ('muskingum',))
        self._plan.muskingum_route(state, routed, num_output_steps, num_routing_per_output)
        self.channel_state = state
        return routed

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

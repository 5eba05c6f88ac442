"""
`Configs`: the reference's configuration object (river_route/routers/Config.py:17-183), key for key, with the
same defaults, the same coercions and the same exceptions, so YAML/JSON files written for the reference load
unchanged.  Engine-only extras (`device`) default to the reference behaviour.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field, fields
from pathlib import Path
from typing import Any, ClassVar

__all__ = ['Configs']

_PATHISH = (str, Path)

# which keys hold paths, and what must exist for each (Config.py:67-77, 112-183)
_INPUT_FILES = ('params_file', 'channel_state_init_file', 'grid_weights_file', 'uh_kernel_file',
                'uh_state_init_file')
_OUTPUT_FILES = ('channel_state_final_file', 'uh_state_final_file')
_OUTPUT_DIRS = ('discharge_dir',)
_INPUT_FILE_LISTS = ('qlateral_files', 'grid_runoff_files')
_OUTPUT_FILE_LISTS = ('discharge_files',)

_CHOICES = {
    'grid_accumulation_type': ('incremental', 'cumulative'),
    'runoff_processing_mode': ('sequential', 'ensemble'),
    'log_level': ('DEBUG', 'INFO', 'PROGRESS', 'WARNING', 'ERROR', 'CRITICAL'),
}


@dataclass
class Configs:
    # core routing files
    params_file: Any = None
    discharge_dir: Any = None
    discharge_files: Any = field(default_factory=list)
    channel_state_init_file: Any = None
    channel_state_final_file: Any = None

    # time options (seconds)
    dt_routing: int = 0
    dt_total: int = 0
    dt_discharge: int = 0
    dt_runoff: int = 0
    start_datetime: str = '1970-01-01'

    # lateral inflow / runoff transformation
    qlateral_files: Any = field(default_factory=list)
    grid_runoff_files: Any = field(default_factory=list)
    grid_weights_file: Any = None
    grid_accumulation_type: str = 'incremental'
    runoff_processing_mode: str = 'sequential'
    uh_kernel_file: Any = None
    uh_state_init_file: Any = None
    uh_state_final_file: Any = None

    # behaviour
    log: bool = True
    progress_bar: bool = True
    log_level: str = 'PROGRESS'
    log_stream: str = 'stdout'
    log_format: str = '%(levelname)s - %(asctime)s - %(message)s'
    var_river_id: str = 'river_id'
    var_discharge: str = 'Q'
    var_grid_runoff: str = 'ro'
    var_x: str = 'x'
    var_y: str = 'y'
    var_t: str = 'time'

    # engine extension: HIP device ordinal the plan lives on
    device: int = 0

    _ALWAYS_REQUIRED: ClassVar[tuple[str, ...]] = ('params_file',)

    def __setattr__(self, name: str, value: object) -> None:
        if getattr(self, '_frozen', False) and name != '_frozen':
            raise AttributeError(f'Configs is frozen — cannot set {name!r}')
        allowed = _CHOICES.get(name)
        if allowed is not None and value not in allowed:
            raise ValueError(f'{name} must be one of {sorted(allowed)}, got {value!r}')
        object.__setattr__(self, name, value)

    def __post_init__(self) -> None:
        self.progress_bar = bool(self.log) and bool(self.progress_bar)
        # a single path given for a list-of-paths key becomes a one-element list
        for key in _INPUT_FILE_LISTS + _OUTPUT_FILE_LISTS:
            val = getattr(self, key)
            if isinstance(val, _PATHISH) and val:
                setattr(self, key, [str(val)])
        for key in _INPUT_FILES + _OUTPUT_FILES + _OUTPUT_DIRS:
            val = getattr(self, key)
            if val:
                setattr(self, key, os.path.abspath(val))
        for key in _INPUT_FILE_LISTS + _OUTPUT_FILE_LISTS:
            val = getattr(self, key)
            if val:
                setattr(self, key, [os.path.abspath(p) for p in val])
        self._resolve_discharge_dir()
        self._verify_inputs_exist()
        self._verify_output_locations_exist()
        for key in self._ALWAYS_REQUIRED:
            if getattr(self, key) in (None, '', []):
                raise ValueError(f'Missing required config: {key}')

    def _resolve_discharge_dir(self) -> None:
        """discharge_dir and discharge_files are exclusive; the directory form names one output per input."""
        if not self.discharge_dir:
            if not self.discharge_files:
                raise ValueError('Provide discharge_dir (or discharge_files for explicit output paths)')
            return
        if self.discharge_files:
            raise ValueError('Provide discharge_dir or discharge_files, not both')
        inputs = self.qlateral_files or self.grid_runoff_files or []
        if inputs:
            self.discharge_files = [os.path.join(self.discharge_dir, f'discharge_{os.path.basename(f)}')
                                    for f in inputs]
        else:
            self.discharge_files = [os.path.join(self.discharge_dir, 'discharge.nc')]

    def _verify_inputs_exist(self) -> None:
        for key in _INPUT_FILES:
            val = getattr(self, key, None)
            if val and not os.path.exists(val):
                raise FileNotFoundError(f'{key} not found: {val}')
        for key in _INPUT_FILE_LISTS:
            for path in getattr(self, key, None) or []:
                if not os.path.exists(path):
                    raise FileNotFoundError(f'{key}: {path} not found')

    def _verify_output_locations_exist(self) -> None:
        targets = [getattr(self, key) for key in _OUTPUT_FILES if getattr(self, key, None)]
        for key in _OUTPUT_FILE_LISTS:
            targets.extend(getattr(self, key, None) or [])
        for path in targets:
            if not os.path.exists(os.path.dirname(path)):
                raise NotADirectoryError(f'Output directory not found for specified output path: {path}')
        for key in _OUTPUT_DIRS:
            val = getattr(self, key, None)
            if val and not os.path.isdir(val):
                raise NotADirectoryError(f'Output directory not found: {val}')

    def deep_validate(self) -> 'Configs':
        """Content checks of the files the configs point at (river_route/routers/Config.py:185-280); like the
        reference's, this is an explicit extra step -- the routers never call it.  Raises ValueError with the
        reference's messages; returns self."""
        import numpy as np
        import pandas as pd

        def need(ok, message):
            if not ok:
                raise ValueError(message)

        pf = self.params_file
        try:
            params = pd.read_parquet(pf)
        except Exception as e:
            raise ValueError('Error reading params file. Must be valid parquet file') from e
        for col in ('river_id', 'downstream_river_id', 'k', 'x'):
            need(col in params.columns, f'{pf} missing {col} column')
        for col in ('river_id', 'downstream_river_id'):
            need(not np.any(params[col].isnull()), f'{pf} {col} column contains null values')
            need(pd.api.types.is_integer_dtype(params[col]), f'{pf} {col} column must be integer type')
            if col == 'river_id':
                need(params[col].is_unique, f'{pf} river_id column must be unique')
        need(not np.any(params['downstream_river_id'] < -1), f'{pf} downstream_river_id column must be -1 or positive integers')
        river_ids = set(params['river_id'].unique())
        need(set(params['downstream_river_id'].unique()).issubset(river_ids | {-1}),
             f'{pf} downstream_river_id values must exist in river_id (except -1)')
        need(not np.any(params['k'] <= 0), f'{pf} k column must be positive')
        need(not (np.any(params['x'] < 0) or np.any(params['x'] > 0.5)), f'{pf} x column must be in the range [0, 0.5]')
        rid = params['river_id'].to_numpy()
        did = params['downstream_river_id'].to_numpy()
        order = np.argsort(rid, kind='stable')
        has = did >= 0
        down_idx = order[np.searchsorted(rid[order], did[has])]
        need(np.all(down_idx > np.flatnonzero(has)), f'{pf} is not topologically sorted (upstream to downstream)')

        if self.grid_weights_file:
            try:
                import xarray as xr
                ds = xr.load_dataset(self.grid_weights_file)
            except Exception as e:
                raise ValueError('Error reading grid weights file. Must be valid netCDF file') from e
            names = ('river_id', 'x_index', 'y_index', 'x', 'y', 'area_sqm', 'proportion')
            for v in names:
                need(v in ds, f'Grid weights file missing {v} variable')
            need(not np.any(ds['river_id'].isnull()), 'Grid weights river_id variable contains null values')
            need(pd.api.types.is_integer_dtype(ds['river_id'].dtype), 'Grid weights river_id variable must be integer type')
            need(set(ds['river_id'].values).issubset(river_ids), 'Grid weights river_id values must exist in params river_id')
            for v in names[1:]:
                need(not np.any(ds[v].isnull()), f'Grid weights {v} variable contains null values')
                need(pd.api.types.is_numeric_dtype(ds[v].dtype), f'Grid weights {v} variable must be numeric type')
            need(not np.any(ds['area_sqm'] <= 0), 'Grid weights area_sqm variable must be positive')
            need(not (np.any(ds['proportion'] <= 0) or np.any(ds['proportion'] > 1)),
                 'Grid weights proportion variable must be in the range (0, 1]')
            need(np.allclose(ds['proportion'].groupby(ds['river_id']).sum().values, 1.0),
                 'Grid weights proportion variable must sum to 1 for each river_id')

        if self.channel_state_init_file:
            try:
                state = pd.read_parquet(self.channel_state_init_file)
            except Exception as e:
                raise ValueError('Error reading initial state file. Must be valid parquet file') from e
            need('Q' in state.columns, 'Initial state file missing Q column')
            need(not np.any(state['Q'].isnull()), 'Initial state file Q column contains null values')
            need(pd.api.types.is_numeric_dtype(state['Q']), 'Initial state file Q column must be numeric type')
            need(not np.any(state['Q'] < 0), 'Initial state file Q column must be non-negative')
            need(state.shape[0] == params.shape[0], f'Initial state file must have the same number of rows as {pf}')
        return self

    def as_dict(self) -> dict[str, Any]:
        return {f.name: getattr(self, f.name) for f in fields(self)}

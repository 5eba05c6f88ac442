"""
Gridded runoff -> catchment lateral inflow: the step immediately upstream of the routing hot path
(river_route/runoff.py:218-352).  Same function name, keyword arguments, order of
operations and units as the reference; the weights product, the cumulative difference, the clip, the NaN fill and
the area scaling run on the GPU (`rr_runoff_to_qlateral`, river_route_amd/csrc/rr_kernels_runoff.hpp:
k_runoff_to_qlateral), the index bookkeeping (pandas) and the rare irregular-time-step resampling stay on the host
exactly as the reference does them.

The reference returns an xarray Dataset; xarray is not a dependency of this package, so the result is a small
`QlateralDataset` with the same names: `ds['qlateral'].values` (time, river_id), `ds['time'].values`,
`ds['river_id'].values`, `ds.attrs`, `'qlateral' in ds`, `ds.dims`.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass, field

import numpy as np

from . import engine
from .io import _decode_cf_time, read_variables

__all__ = ['runoff_to_qlateral', 'QlateralDataset', 'RunoffSource', 'prepare_runoff']

logger = logging.getLogger(__name__)


@dataclass
class _Var:
    values: np.ndarray
    dims: tuple
    attrs: dict = field(default_factory=dict)


class QlateralDataset:
    """The three variables runoff_to_qlateral returns (river_route/runoff.py:343-352), by name."""

    def __init__(self, time, river_id, qlateral, units, long_name, timestep, attrs):
        self._vars = {
            'time': _Var(time, ('time',), {'long_name': 'time', 'standard_name': 'time', 'axis': 'T', 'time_step': f'{timestep}'}),
            'river_id': _Var(river_id, ('river_id',), {'long_name': 'unique ID number for each river'}),
            'qlateral': _Var(qlateral, ('time', 'river_id'), {'units': units, 'long_name': long_name}),
        }
        self.dims = {'time': time.shape[0], 'river_id': river_id.shape[0]}
        self.attrs = attrs

    def __getitem__(self, name):
        return self._vars[name]

    def __contains__(self, name):
        return name in self._vars


def _get_conversion_factor(unit):
    """river_route/runoff.py:206-215."""
    if unit is None:
        logger.warning('No units attribute found. Assuming meters')
        return 1
    if unit in ('m', 'meters', 'kg m-2'):
        return 1
    if unit in ('mm', 'millimeters'):
        return .001
    raise ValueError(f'Unknown units: {unit}')


def _read_runoff_points(paths, var_runoff, var_x, var_y, var_t, x_index, y_index):
    """(T, points) block of the runoff variable at the (x, y) index pairs, concatenated over the files in time
    order, its `units` attribute and the time axis (runoff.py:266-279: open_mfdataset + isel + transpose)."""
    blocks, times, units = [], [], None
    for path in paths:
        got = read_variables(path, [var_runoff, var_t])
        arr, dims, attrs = got[var_runoff]
        for d in (var_t, var_x, var_y):
            if d not in dims:
                raise KeyError(f'{var_runoff} in {path} has no dimension {d!r} (has {dims})')
        arr = np.moveaxis(arr, [dims.index(var_t), dims.index(var_y), dims.index(var_x)], [0, 1, 2])
        if arr.ndim != 3:
            raise ValueError(f'{var_runoff} in {path} must have exactly the dimensions ({var_t}, {var_y}, {var_x})')
        blocks.append(arr[:, y_index, x_index])
        units = units if units is not None else attrs.get('units')
        tv, _, tattrs = got[var_t]
        if np.issubdtype(tv.dtype, np.datetime64):
            times.append(tv.astype('datetime64[s]'))
        else:
            times.append(_decode_cf_time(tv, tattrs.get('units', 'seconds since 1970-01-01')))
    time_index = np.concatenate(times)
    block = np.concatenate(blocks, axis=0)
    if len(paths) > 1:
        order = np.argsort(time_index, kind='stable')      # open_mfdataset combines by coordinates
        time_index, block = time_index[order], block[order]
    return block, units, time_index


@dataclass
class RunoffSource:
    """Everything rr_runoff_to_qlateral needs for one set of runoff files, before any arithmetic of the path: the
    (time, points) runoff block, the CSR weights (proportion x unit conversion, duplicates summed, ascending point order
    as the reference's csr_matrix), catchment areas, flags, the time axis and the river ids in column order.  The routers
    upload it as it is, so that the catchment inflow is computed on the GPU and routed from there without a trip through host
    memory (TransformMuskingum._router_device_runoff; rr_rapid_route_runoff_dev takes the same pieces and computes the inflow
    on the way into the engine's records); `to_array()` is the reference's (time, river) array."""
    runoff_tp: np.ndarray
    indptr: np.ndarray
    indices: np.ndarray
    weights: np.ndarray
    area: np.ndarray
    flags: int
    time_index: np.ndarray
    river_ids: np.ndarray
    irregular: bool
    device: int = 0

    def point_major(self):
        """(points, padded time) copy of the block, rows padded to whole 16-step chunks, float32 or float64 as read."""
        block = self.runoff_tp if self.runoff_tp.dtype in (np.float32, np.float64) else self.runoff_tp.astype(np.float64)
        T, n_points = block.shape
        t_pad = -(-T // 16) * 16
        out = np.zeros((n_points, t_pad), dtype=block.dtype)
        out[:, :T] = block.T
        return out

    def to_array(self, as_volumes: bool, keep_nan: bool = False) -> np.ndarray:
        return engine.runoff_to_qlateral(self.indptr, self.indices, self.weights, self.runoff_tp, self.area if as_volumes else None,
                                         self.flags | (engine.RUNOFF_KEEP_NAN if keep_nan else 0), self.device)


def prepare_runoff(runoff_data, grid_weights_file, *, var_runoff: str = 'ro', var_x: str = 'lon', var_y: str = 'lat', var_t: str = 'time',
                   var_river_id: str = 'river_id', runoff_depth_unit: str | None = None, cumulative: bool = False,
                   force_positive_runoff: bool = False, force_uniform_timesteps: bool = True, device: int = 0) -> RunoffSource:
    """File reading and index bookkeeping of runoff_to_qlateral (river_route/runoff.py:255-298), no arithmetic of the path."""
    import pandas as pd
    import scipy.sparse

    cols = [var_river_id, 'x_index', 'y_index', 'proportion', 'area_sqm']
    table = read_variables(grid_weights_file, cols)
    weight_df = pd.DataFrame({c: np.asarray(table[c][0]).ravel() for c in cols})
    unique_indexes = (weight_df[['x_index', 'y_index']].drop_duplicates().reset_index(drop=True).reset_index().astype(int))
    river_ids_ordered = weight_df[var_river_id].drop_duplicates().to_numpy()     # index already topologically sorted

    paths = [runoff_data] if isinstance(runoff_data, (str, bytes)) or hasattr(runoff_data, '__fspath__') else list(runoff_data)
    runoff_raw, file_units, time_index = _read_runoff_points(
        paths, var_runoff, var_x, var_y, var_t, unique_indexes['x_index'].to_numpy(), unique_indexes['y_index'].to_numpy())
    conversion_factor = _get_conversion_factor(runoff_depth_unit or (file_units if file_units is not None else 'm'))

    # sparse weights (n_rivers, n_unique_points), built the way the reference builds them so that duplicate entries
    # are summed and the terms of a row are stored in the same (ascending point) order
    point_idx = weight_df[['x_index', 'y_index']].merge(unique_indexes, on=['x_index', 'y_index'], how='left')['index'].to_numpy()
    river_id_to_row = pd.Series(np.arange(len(river_ids_ordered)), index=river_ids_ordered)
    river_idx = river_id_to_row.loc[weight_df[var_river_id].to_numpy()].to_numpy()
    weights = scipy.sparse.csr_matrix((weight_df['proportion'].to_numpy() * conversion_factor, (river_idx, point_idx)),
                                      shape=(len(river_ids_ordered), len(unique_indexes)))
    weights.sum_duplicates()
    catchment_area = weight_df.groupby(var_river_id)['area_sqm'].sum().reindex(river_ids_ordered).to_numpy()

    time_diff = np.diff(time_index)
    irregular = bool(time_index.shape[0] > 2 and not np.all(time_diff == time_index[1] - time_index[0]) and force_uniform_timesteps)
    flags = (engine.RUNOFF_CUMULATIVE if cumulative else 0) | (engine.RUNOFF_FORCE_POSITIVE if force_positive_runoff else 0)
    return RunoffSource(runoff_raw, weights.indptr.astype(np.int32), weights.indices.astype(np.int32), np.ascontiguousarray(weights.data, dtype=np.float64),
                        np.ascontiguousarray(catchment_area, dtype=np.float64), flags, time_index, river_ids_ordered.astype(np.int64, copy=False),
                        irregular, device)


def runoff_to_qlateral(runoff_data, grid_weights_file, *, var_runoff: str = 'ro', var_x: str = 'lon', var_y: str = 'lat',
                       var_t: str = 'time', var_river_id: str = 'river_id', runoff_depth_unit: str | None = None,
                       cumulative: bool = False, force_positive_runoff: bool = False,
                       force_uniform_timesteps: bool = True, as_volumes: bool = False, device: int = 0) -> QlateralDataset:
    """Area-weighted aggregation of gridded runoff depths to per-catchment lateral inflow (depths in m, or volumes
    in m3 with `as_volumes`), river_route/runoff.py:218-352.  `device` (HIP ordinal) is the only extra argument."""
    import pandas as pd
    src = prepare_runoff(runoff_data, grid_weights_file, var_runoff=var_runoff, var_x=var_x, var_y=var_y, var_t=var_t,
                         var_river_id=var_river_id, runoff_depth_unit=runoff_depth_unit, cumulative=cumulative,
                         force_positive_runoff=force_positive_runoff, force_uniform_timesteps=force_uniform_timesteps, device=device)
    time_index, river_ids_ordered, catchment_area = src.time_index, src.river_ids, src.area
    if not src.irregular:
        qlateral = src.to_array(as_volumes)
    else:
        # runoff.py:311-330: the resampling sits between the clip and the NaN fill, so the device stops before the fill
        qlateral = src.to_array(False, keep_nan=True)
        timestep = int((time_index[1] - time_index[0]) / np.timedelta64(1, 's'))
        logger.warning(f'Time steps are not uniform, resampling to the first timestep: {timestep} seconds')
        df = pd.DataFrame(qlateral, index=time_index, columns=river_ids_ordered)
        df = df.cumsum().resample(rule=f'{timestep}s').interpolate(method='linear')
        df = pd.concat([df.iloc[[0]], df.diff().iloc[1:]])
        time_index = df.index.values.astype('datetime64[s]')
        qlateral = df.to_numpy(dtype=np.float64)
        qlateral[np.isnan(qlateral)] = 0.0
        if as_volumes:
            qlateral *= catchment_area[np.newaxis, :]

    units, long_name = ('m3', 'Incremental qlateral volumes') if as_volumes else ('m', 'Incremental qlateral depths')
    start_date = pd.Timestamp(time_index[0]).strftime('%Y%m%d%H')
    end_date = pd.Timestamp(time_index[-1]).strftime('%Y%m%d%H')
    timestep = int((time_index[1] - time_index[0]) / np.timedelta64(1, 's')) if len(time_index) > 1 else 0
    attrs = {'title': f'Incremental qlateral {long_name.split()[-1]}',
             'description': f'Incremental qlateral ({units}) for each river', 'source': 'river_route_amd',
             'history': f'Created on {pd.Timestamp.now().strftime("%Y-%m-%d %H:%M:%S")}',
             'start_date': start_date, 'end_date': end_date}
    return QlateralDataset(time_index.astype('datetime64[s]'), river_ids_ordered.astype(np.int64, copy=False), qlateral, units,
                           long_name, timestep, attrs)

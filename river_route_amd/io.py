"""
File I/O either side of the hot path, kept minimal: lateral-inflow netCDF in (io-file-schema of the reference:
dims time x river_id, variable `qlateral`), discharge netCDF out (river_route/routers/Muskingum.py:337-351).
netCDF4 / xarray are used when installed; otherwise scipy's NetCDF-3 reader/writer.
"""
from __future__ import annotations

import re

import numpy as np

__all__ = ['read_qlateral', 'write_discharge', 'read_variables']

_UNIT_SECONDS = {'second': 1, 'seconds': 1, 'sec': 1, 'secs': 1, 's': 1, 'minute': 60, 'minutes': 60, 'min': 60,
                 'hour': 3600, 'hours': 3600, 'h': 3600, 'hr': 3600, 'day': 86400, 'days': 86400, 'd': 86400}


def _decode_cf_time(values: np.ndarray, units: str) -> np.ndarray:
    m = re.match(r'\s*(\w+)\s+since\s+(.+?)\s*$', units)
    if not m or m.group(1).lower() not in _UNIT_SECONDS:
        raise ValueError(f'cannot decode time units {units!r}')
    origin = np.datetime64(m.group(2).strip().replace(' ', 'T').rstrip('Z'), 's')
    secs = np.round(np.asarray(values, dtype=np.float64) * _UNIT_SECONDS[m.group(1).lower()]).astype(np.int64)
    return origin + secs.astype('timedelta64[s]')


def read_qlateral(path, var_t: str = 'time', var: str = 'qlateral', keep_float32: bool = False):
    """-> (dates datetime64[s][T], array float64[T, n]) as TransformMuskingum._qlateral_generator yields them
    (river_route/routers/TransformMuskingum.py:33-36).  keep_float32: a variable the file stores as float32 is returned as it
    is (the caller converts on the device: half the bytes to upload and to read, the same float64 values)."""
    def cast(values):
        values = np.asarray(values)
        if keep_float32 and values.dtype == np.float32:
            return np.ascontiguousarray(values)
        return values.astype(np.float64, copy=False)
    try:
        import xarray as xr
        with xr.open_dataset(path) as ds:
            return ds[var_t].values.astype('datetime64[s]'), cast(ds[var].values)
    except ImportError:
        pass
    try:
        import netCDF4 as nc
        with nc.Dataset(str(path)) as ds:
            tv = ds[var_t]
            dates = _decode_cf_time(np.asarray(tv[:]), tv.units)
            return dates, cast(np.asarray(ds[var][:]))
    except ImportError:
        pass
    from scipy.io import netcdf_file
    with netcdf_file(str(path), 'r', mmap=False) as ds:
        tv = ds.variables[var_t]
        units = tv.units.decode() if isinstance(tv.units, bytes) else tv.units
        dates = _decode_cf_time(tv[:].copy(), units)
        raw = ds.variables[var][:]
        return dates, cast(np.array(raw, dtype=raw.dtype.newbyteorder('=')))


def _cf_decode(raw: np.ndarray, attrs: dict) -> np.ndarray:
    """CF conventions as xarray's mask_and_scale applies them (the reference reads runoff through xarray.open_mfdataset,
    river_route/runoff.py:255-270): cells equal to _FillValue / missing_value (or outside valid_min / valid_max /
    valid_range) become NaN -- compared on the RAW stored values -- then scale_factor / add_offset unpack the rest."""
    mask = None
    for key in ('_FillValue', 'missing_value'):
        if key in attrs:
            for fill in np.atleast_1d(attrs[key]):
                hit = np.isnan(raw) if isinstance(fill, (float, np.floating)) and np.isnan(fill) else raw == fill
                mask = hit if mask is None else (mask | hit)
    lo, hi = attrs.get('valid_min'), attrs.get('valid_max')
    if 'valid_range' in attrs:
        lo, hi = np.atleast_1d(attrs['valid_range'])[:2]
    if lo is not None:
        mask = (raw < lo) if mask is None else (mask | (raw < lo))
    if hi is not None:
        mask = (raw > hi) if mask is None else (mask | (raw > hi))
    packed = 'scale_factor' in attrs or 'add_offset' in attrs
    if mask is None and not packed:
        return raw
    out = raw.astype(np.float64) if (packed or not np.issubdtype(raw.dtype, np.floating)) else raw.astype(raw.dtype, copy=True)
    if packed:
        out = out * np.float64(attrs.get('scale_factor', 1.0)) + np.float64(attrs.get('add_offset', 0.0))
    if mask is not None and mask.any():
        out[mask] = np.nan
    return out


def read_variables(path, names):
    """-> {name: (array, dims tuple, attrs dict)} for the named variables of one netCDF file, fill values masked to NaN and
    packed values unpacked (time variables stay raw numbers: decode them with the `units` attribute).  Same backends as
    read_qlateral, minus xarray's time decoding."""
    out = {}
    try:
        import netCDF4 as nc
        with nc.Dataset(str(path)) as ds:
            ds.set_auto_maskandscale(False)       # raw values: one decoder for both backends
            for name in names:
                v = ds[name]
                attrs = {k: v.getncattr(k) for k in v.ncattrs()}
                out[name] = (_cf_decode(np.asarray(v[:]), attrs), tuple(v.dimensions), attrs)
        return out
    except ImportError:
        pass
    from scipy.io import netcdf_file
    with netcdf_file(str(path), 'r', mmap=False, maskandscale=False) as ds:
        for name in names:
            if name not in ds.variables:
                raise KeyError(f'{name} not in {path}')
            v = ds.variables[name]
            attrs = {k: (a.decode() if isinstance(a, bytes) else a) for k, a in v._attributes.items()}
            arr = np.array(v[:])
            arr = arr.astype(arr.dtype.newbyteorder('='), copy=False)      # NetCDF-3 is big-endian on disk
            out[name] = (_cf_decode(arr, attrs), tuple(v.dimensions), attrs)
    return out


def write_discharge(path, dates, q_array, river_ids, var_river_id='river_id', var_discharge='Q', routed_file=''):
    """Discharge file with the reference's layout: dims (time, river_id); time f8 'seconds since <first date>';
    ids i4; Q f4 with long_name / standard_name / aggregation_method / units attributes."""
    dates = np.asarray(dates).astype('datetime64[s]')
    origin = str(dates[0]).replace('T', ' ')
    seconds = (dates - dates[0]).astype('timedelta64[s]').astype(np.int64)
    attrs = dict(long_name='Discharge at catchment outlet', standard_name='discharge',
                 aggregation_method='mean', units='m3 s-1')
    try:
        import netCDF4 as nc
        with nc.Dataset(str(path), mode='w', format='NETCDF4') as ds:
            ds.createDimension('time', size=q_array.shape[0])
            ds.createDimension(var_river_id, size=q_array.shape[1])
            ds.runoff_file = str(routed_file)
            tv = ds.createVariable('time', 'f8', ('time',))
            tv.units = f'seconds since {origin}'
            tv[:] = seconds
            iv = ds.createVariable(var_river_id, 'i4', (var_river_id,))
            iv[:] = river_ids
            fv = ds.createVariable(var_discharge, 'f4', ('time', var_river_id))
            fv[:] = q_array
            for k, v in attrs.items():
                setattr(fv, k, v)
        return
    except ImportError:
        pass
    from scipy.io import netcdf_file
    with netcdf_file(str(path), 'w', version=2) as ds:
        # a fixed-size NetCDF-3 variable holds less than 2 GiB (its size field is 32 bits): beyond that `time` is the record
        # dimension (same dimensions and variables to any reader; 1M reaches x 744 rows of float32 are 2.98 GB)
        big = q_array.shape[0] * q_array.shape[1] * 4 >= (1 << 31) - 4
        ds.createDimension('time', None if big else q_array.shape[0])
        ds.createDimension(var_river_id, q_array.shape[1])
        ds.runoff_file = str(routed_file)
        tv = ds.createVariable('time', 'f8', ('time',))
        tv.units = f'seconds since {origin}'
        tv[:] = seconds.astype(np.float64)
        iv = ds.createVariable(var_river_id, 'i4', (var_river_id,))
        iv[:] = np.asarray(river_ids).astype(np.int32)
        fv = ds.createVariable(var_discharge, 'f4', ('time', var_river_id))
        fv[:] = np.asarray(q_array, dtype=np.float32)
        for k, v in attrs.items():
            setattr(fv, k, v)

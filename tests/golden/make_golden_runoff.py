"""
Generates tests/golden/runoff.npz by RUNNING THE REFERENCE's runoff_to_qlateral (river_route/runoff.py:218-373,
read-only at /root/reference) on small seeded weight tables and runoff grids.

Runs only in the build container (it needs /root/reference).  The reference function executes as it is written --
the sparse weights product, the cumulative difference, the clip, the pandas resampling of irregular time steps, the
NaN fill and the area scaling are all its own statements running on the installed numpy / scipy / pandas.  What is
NOT installed is xarray (and geopandas / shapely, which runoff.py imports for its GIS functions): geopandas and
shapely are empty placeholder modules nothing on this path touches, and xarray is replaced by the small FILE-ACCESS
stand-in below (`_Dataset` / `_DataArray` / `_open` / `_open_mf`): it opens NetCDF-3 files with scipy, decodes CF time, and provides the container
calls the function makes (`ds[[...]].to_dataframe()`, `ds[var].isel(...).transpose(...).values`, `ds[var].attrs`,
`ds[var].to_numpy()`, and the `xr.Dataset` / `xr.DataArray` constructors of the return value).  No arithmetic of the
path runs in the stand-in.  Only inputs and the outputs the reference produced are written to the .npz.

    python tests/golden/make_golden_runoff.py
"""
import importlib
import os
import re
import sys
import tempfile
import types
import typing

import numpy as np
import pandas as pd

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, 'tests'))


# ---------------------------------------------------------------- file-access stand-in for xarray (see docstring)
class _DataArray:
    def __init__(self, data, dims=None, attrs=None, name=None):
        self.values = np.asarray(data)
        self.dims = (dims,) if isinstance(dims, str) else tuple(dims or ())
        self.attrs = dict(attrs or {})

    def to_numpy(self):
        return self.values

    def isel(self, indexers):
        # pointwise (vectorised) indexing: every indexer is a DataArray over the same new dimension
        new_dim = next(iter(indexers.values())).dims[0]
        axes = [self.dims.index(d) for d in indexers]
        idx = [slice(None)] * self.values.ndim
        for d, ix in indexers.items():
            idx[self.dims.index(d)] = np.asarray(ix.values)
        out = self.values[tuple(idx)]
        # numpy puts the broadcast dimension first when the indexed axes are not adjacent, else in place of the first
        adjacent = sorted(axes) == list(range(min(axes), max(axes) + 1))
        rest = [d for d in self.dims if d not in indexers]
        if adjacent:
            pos = min(axes)
            dims = rest[:pos] + [new_dim] + rest[pos:]
        else:
            dims = [new_dim] + rest
        return _DataArray(out, dims, self.attrs)

    def transpose(self, *dims):
        return _DataArray(np.transpose(self.values, [self.dims.index(d) for d in dims]), dims, self.attrs)


class _Dataset:
    def __init__(self, data_vars=None, coords=None, attrs=None):
        self.vars = {}
        for group in (data_vars or {}), (coords or {}):
            for k, v in group.items():
                self.vars[k] = v
        self.attrs = dict(attrs or {})

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def __contains__(self, k):
        return k in self.vars

    @property
    def dims(self):
        d = {}
        for v in self.vars.values():
            d.update(dict(zip(v.dims, v.values.shape)))
        return d

    def __getitem__(self, key):
        if isinstance(key, list):
            return _Dataset({k: self.vars[k] for k in key})
        return self.vars[key]

    def to_dataframe(self):
        return pd.DataFrame({k: v.values for k, v in self.vars.items()})


def _open(path):
    from scipy.io import netcdf_file
    out = {}
    with netcdf_file(str(path), 'r', mmap=False) as ds:
        for name, v in ds.variables.items():
            attrs = {k: (a.decode() if isinstance(a, bytes) else a) for k, a in v._attributes.items()}
            arr = np.array(v[:])
            arr = arr.astype(arr.dtype.newbyteorder('='), copy=False)
            m = re.match(r'\s*(\w+)\s+since\s+(.+?)\s*$', str(attrs.get('units', '')))
            if m:       # CF time, decoded as xarray decodes it
                unit = {'seconds': 1, 'hours': 3600, 'days': 86400}[m.group(1)]
                origin = np.datetime64(m.group(2).replace(' ', 'T'), 's')
                arr = origin + np.round(arr.astype(np.float64) * unit).astype(np.int64).astype('timedelta64[s]')
                arr = arr.astype('datetime64[ns]')
            out[name] = _DataArray(arr, v.dimensions, attrs)
    return _Dataset(out)


def _open_mf(paths):
    paths = [paths] if isinstance(paths, (str, os.PathLike)) else list(paths)
    parts = [_open(p) for p in paths]
    if len(parts) == 1:
        return parts[0]
    tdim = next(d for d in parts[0].vars if np.issubdtype(parts[0].vars[d].values.dtype, np.datetime64))
    order = np.argsort([p.vars[tdim].values[0] for p in parts])
    merged = {}
    for name, v in parts[0].vars.items():
        if tdim in v.dims:
            ax = v.dims.index(tdim)
            merged[name] = _DataArray(np.concatenate([parts[i].vars[name].values for i in order], axis=ax), v.dims, v.attrs)
        else:
            merged[name] = v
    return _Dataset(merged)


def load_reference_runoff():
    if not os.path.isdir(os.path.join(REF, 'river_route')):
        raise SystemExit('make_golden_runoff.py: /root/reference is not present; golden vectors can only be '
                         'regenerated in the build container')
    import typing_extensions
    if not hasattr(typing, 'Self'):
        typing.Self = typing_extensions.Self
    for name in ('geopandas', 'shapely', 'shapely.geometry', 'shapely.ops'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['geopandas'].GeoDataFrame = object
    xr = types.ModuleType('xarray')
    xr.Dataset, xr.DataArray, xr.open_dataset, xr.open_mfdataset = _Dataset, _DataArray, _open, _open_mf
    sys.modules['xarray'] = xr
    shell = types.ModuleType('river_route')
    shell.__path__ = [os.path.join(REF, 'river_route')]
    sys.modules['river_route'] = shell
    sys.dont_write_bytecode = True
    return importlib.import_module('river_route.runoff')


def main():
    ref = load_reference_runoff()
    from test_runoff import make_grid_case, GOLDEN_CASES as CASES     # the seeded weight tables / grids of the CPU tests
    from pathlib import Path
    out = {}
    for i, (tag, gk, rk) in enumerate(CASES):
        with tempfile.TemporaryDirectory() as tmp:
            rng = np.random.default_rng(100 + i)
            gk = dict(gk)
            if 'hours' in gk:
                gk['hours'] = np.asarray(gk['hours'])
            wfile, paths, tab, grid, hours = make_grid_case(Path(tmp), rng, **gk)
            ds = ref.runoff_to_qlateral(paths if len(paths) > 1 else paths[0], wfile, var_x='longitude', var_y='latitude',
                                        var_t='valid_time', **rk)
            out[f'{tag}/table'] = tab
            out[f'{tag}/grid'] = grid
            out[f'{tag}/hours'] = np.asarray(hours, dtype=np.int64)
            out[f'{tag}/qlateral'] = np.asarray(ds['qlateral'].values, dtype=np.float64)
            out[f'{tag}/river_id'] = np.asarray(ds['river_id'].values, dtype=np.int64)
            out[f'{tag}/time'] = np.asarray(ds['time'].values).astype('datetime64[s]').astype(np.int64)
            out[f'{tag}/units'] = np.array(ds['qlateral'].attrs['units'])
    np.savez_compressed(os.path.join(HERE, 'runoff.npz'), **out)
    print('wrote runoff.npz:', {k: v.shape for k, v in out.items() if k.endswith('qlateral')})


if __name__ == '__main__':
    main()

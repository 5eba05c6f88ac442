"""Gridded runoff -> catchment inflow (river_route/runoff.py:218-352).

Pinned to the reference: tests/golden/runoff.npz holds the outputs of the reference's own runoff_to_qlateral for five
seeded cases (tests/golden/make_golden_runoff.py); `test_against_reference_golden` reproduces them with the host
logic of river_route_amd.runoff + the oracle's arithmetic (CPU) and + the HIP kernel (`-m gpu`).  The other CPU
tests check the oracle against an independent dense evaluation and the host logic against a table-driven brute
force.  tests/test_gpu_runoff.py compares the kernel alone with the oracle."""
import os
import sys

import numpy as np
import pytest
import scipy.sparse

from conftest import REPO, assert_close

from oracle import oracle  # noqa: E402


def random_weights(rng, n_rivers, n_points, max_nnz=6, empty_every=17):
    rows, cols, data = [], [], []
    for r in range(n_rivers):
        k = 0 if (empty_every and r % empty_every == 3) else int(rng.integers(1, min(max_nnz, n_points) + 1))
        c = rng.choice(n_points, size=k, replace=False)
        w = rng.random(k)
        w = w / w.sum() if k else w
        rows += [r] * k
        cols += list(c)
        data += list(w)
    W = scipy.sparse.csr_matrix((np.array(data), (np.array(rows, dtype=np.int64), np.array(cols, dtype=np.int64))),
                                shape=(n_rivers, n_points))
    W.sum_duplicates()
    return W


def dense_reference(W, runoff, area, cumulative, clip, keep_nan=False):
    """Independent evaluation: dense matrix, python loop over time."""
    D = W.toarray()
    T = runoff.shape[0]
    S = np.empty((T, W.shape[0]))
    for t in range(T):
        S[t] = [sum(D[r, p] * float(runoff[t, p]) for p in np.nonzero(D[r])[0]) for r in range(W.shape[0])]
    q = S.copy()
    if cumulative:
        q[1:] = S[1:] - S[:-1]
    if clip:
        q = np.where(q < 0, 0.0, q)
    if not keep_nan:
        q = np.where(np.isnan(q), 0.0, q)
    return q * area[None, :] if area is not None else q


@pytest.mark.parametrize('cumulative,clip,volumes,dtype', [(False, False, False, np.float64), (True, False, True, np.float32),
                                                           (False, True, True, np.float32), (True, True, False, np.float64)])
def test_oracle_core_vs_dense(cumulative, clip, volumes, dtype):
    rng = np.random.default_rng(5)
    W = random_weights(rng, 40, 25)
    runoff = (rng.random((11, 25)) - 0.2).astype(dtype)
    runoff[3, 4] = np.nan
    if cumulative:
        runoff = np.cumsum(runoff, axis=0).astype(dtype)
    area = rng.uniform(1e6, 5e7, 40) if volumes else None
    got = oracle.runoff_to_qlateral_core(W, runoff, area, cumulative, clip)
    want = dense_reference(W, runoff, area, cumulative, clip)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())


def write_nc3(path, dims, variables):
    """variables: {name: (dims tuple, array, attrs)} -> NetCDF-3 file (the reader every image has)."""
    from scipy.io import netcdf_file
    with netcdf_file(str(path), 'w', version=2) as ds:
        for d, size in dims.items():
            ds.createDimension(d, size)
        for name, (vdims, arr, attrs) in variables.items():
            arr = np.asarray(arr)
            v = ds.createVariable(name, arr.dtype.char if arr.dtype != np.int64 else 'i', vdims)
            v[:] = arr.astype(np.int32) if arr.dtype == np.int64 else arr
            for k, a in attrs.items():
                setattr(v, k, a)


def write_grid_case(tmp_path, tab, grid, hours, units, files=1):
    """Weight table + runoff grid(s) (dims time, lat, lon; NetCDF-3) from arrays -> (weights file, runoff files)."""
    ny, nx = grid.shape[1:]
    wfile = tmp_path / 'weights.nc'
    write_nc3(wfile, {'index': len(tab)}, {
        'river_id': (('index',), tab[:, 0].astype(np.int64), {}), 'x_index': (('index',), tab[:, 1].astype(np.int64), {}),
        'y_index': (('index',), tab[:, 2].astype(np.int64), {}), 'proportion': (('index',), tab[:, 3], {}),
        'area_sqm': (('index',), tab[:, 4], {})})
    paths = []
    per = len(hours) // files
    for f in range(files):
        sl = slice(f * per, (f + 1) * per if f < files - 1 else len(hours))
        p = tmp_path / f'runoff_{f}.nc'
        write_nc3(p, {'valid_time': len(hours[sl]), 'latitude': ny, 'longitude': nx}, {
            'valid_time': (('valid_time',), hours[sl].astype(np.float64), {'units': 'hours since 2020-01-01 00:00:00'}),
            'latitude': (('latitude',), np.linspace(50, 46, ny), {}), 'longitude': (('longitude',), np.linspace(5, 11, nx), {}),
            'ro': (('valid_time', 'latitude', 'longitude'), grid[sl], {'units': units})})
        paths.append(str(p))
    return str(wfile), paths


def make_grid_case(tmp_path, rng, n_rivers=30, nx=7, ny=5, T=10, units='mm', dtype=np.float32, cumulative=False,
                   files=1, hours=None):
    """Weight table with rivers in 'topological' (first-appearance) order, repeated cells and one duplicated entry;
    runoff grid(s) with dims (time, lat, lon)."""
    river_ids = 1000 + rng.permutation(n_rivers) * 3
    rows = []
    for rid in river_ids:
        k = int(rng.integers(1, 5))
        cells = rng.choice(nx * ny, size=k, replace=False)
        prop = rng.random(k)
        prop /= prop.sum()
        for c, p in zip(cells, prop):
            rows.append((rid, c % nx, c // nx, p, float(rng.uniform(1e5, 1e7))))
    rows.append((rows[0][0], rows[0][1], rows[0][2], 0.125, 5e5))        # same river, same cell again
    tab = np.array(rows)
    hours = np.arange(T * files) if hours is None else np.asarray(hours)
    grid = (rng.random((len(hours), ny, nx)) * 3 - 0.3).astype(dtype)
    grid[2, 1, 2] = np.nan
    if cumulative:
        grid = np.cumsum(np.nan_to_num(grid), axis=0).astype(dtype)
    wfile, paths = write_grid_case(tmp_path, tab, grid, hours, units, files)
    return wfile, paths, tab, grid, hours


def brute_force(tab, grid, conv, cumulative, clip, volumes):
    """Per river: sum over its table rows of proportion * conversion * cell series -- no sparse matrix, no pandas."""
    rids = []
    for r in tab[:, 0]:
        if r not in rids:
            rids.append(r)
    T = grid.shape[0]
    S = np.zeros((T, len(rids)))
    area = np.zeros(len(rids))
    for rid, xi, yi, prop, a in tab:
        j = rids.index(rid)
        S[:, j] += prop * conv * grid[:, int(yi), int(xi)].astype(np.float64)
        area[j] += a
    q = S.copy()
    if cumulative:
        q[1:] = S[1:] - S[:-1]
    if clip:
        q = np.where(q < 0, 0.0, q)
    q = np.where(np.isnan(q), 0.0, q)
    return np.array(rids, dtype=np.int64), (q * area[None, :] if volumes else q)


@pytest.fixture
def oracle_device(monkeypatch):
    """river_route_amd.runoff with the device call replaced by the oracle: host logic only, no GPU."""
    from river_route_amd import engine

    def fake(indptr, indices, weights, runoff_tp, area=None, flags=0, device=0):
        W = scipy.sparse.csr_matrix((weights, indices, indptr), shape=(len(indptr) - 1, runoff_tp.shape[1]))
        return oracle.runoff_to_qlateral_core(W, runoff_tp, area, bool(flags & engine.RUNOFF_CUMULATIVE),
                                              bool(flags & engine.RUNOFF_FORCE_POSITIVE), bool(flags & engine.RUNOFF_KEEP_NAN))
    monkeypatch.setattr(engine, 'runoff_to_qlateral', fake)


@pytest.mark.parametrize('units,cumulative,clip,volumes,files', [('mm', False, False, True, 1), ('m', True, True, False, 1),
                                                                 ('kg m-2', False, True, True, 3)])
def test_host_logic_vs_brute_force(tmp_path, oracle_device, units, cumulative, clip, volumes, files):
    from river_route_amd.runoff import runoff_to_qlateral
    rng = np.random.default_rng(11)
    wfile, paths, tab, grid, hours = make_grid_case(tmp_path, rng, units=units, cumulative=cumulative, files=files)
    ds = runoff_to_qlateral(paths if files > 1 else paths[0], wfile, var_x='longitude', var_y='latitude', var_t='valid_time',
                            cumulative=cumulative, force_positive_runoff=clip, as_volumes=volumes)
    rids, want = brute_force(tab, grid, 0.001 if units == 'mm' else 1.0, cumulative, clip, volumes)
    assert 'qlateral' in ds and ds['qlateral'].dims == ('time', 'river_id')
    np.testing.assert_array_equal(ds['river_id'].values, rids)
    np.testing.assert_array_equal(ds['time'].values, np.datetime64('2020-01-01T00:00:00') + hours.astype('timedelta64[h]'))
    np.testing.assert_allclose(ds['qlateral'].values, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
    assert ds['qlateral'].attrs['units'] == ('m3' if volumes else 'm')
    assert ds['time'].attrs['time_step'] == '3600'


def test_unknown_units_and_missing_dimension(tmp_path, oracle_device):
    from river_route_amd.runoff import runoff_to_qlateral
    rng = np.random.default_rng(2)
    wfile, paths, *_ = make_grid_case(tmp_path, rng, units='furlongs')
    with pytest.raises(ValueError, match='Unknown units'):
        runoff_to_qlateral(paths[0], wfile, var_x='longitude', var_y='latitude', var_t='valid_time')
    with pytest.raises(KeyError):
        runoff_to_qlateral(paths[0], wfile, var_x='lon', var_y='latitude', var_t='valid_time', runoff_depth_unit='m')


def test_irregular_time_steps_are_resampled(tmp_path, oracle_device):
    """runoff.py:311-325: cumulate, resample to the first step, interpolate, difference; volumes are conserved."""
    from river_route_amd.runoff import runoff_to_qlateral
    rng = np.random.default_rng(3)
    hours = np.array([0, 1, 2, 4, 5, 6, 9, 10])
    wfile, paths, tab, grid, _ = make_grid_case(tmp_path, rng, units='m', dtype=np.float64, hours=hours)
    ds = runoff_to_qlateral(paths[0], wfile, var_x='longitude', var_y='latitude', var_t='valid_time', as_volumes=True)
    assert ds['time'].values.shape[0] == 11 and ds['time'].attrs['time_step'] == '3600'
    _, irregular_total = brute_force(tab, grid, 1.0, False, False, True)
    np.testing.assert_allclose(ds['qlateral'].values.sum(axis=0), irregular_total.sum(axis=0), rtol=1e-9)
    raw = runoff_to_qlateral(paths[0], wfile, var_x='longitude', var_y='latitude', var_t='valid_time', as_volumes=True,
                             force_uniform_timesteps=False)
    assert raw['time'].values.shape[0] == 8


# ---- pinned to the reference: tests/golden/runoff.npz holds what river_route.runoff.runoff_to_qlateral itself returned
# for these cases (tests/golden/make_golden_runoff.py)
GOLDEN_CASES = [   # tag, make_grid_case kwargs, runoff_to_qlateral kwargs
    ('incr_mm_vol', dict(units='mm'), dict(as_volumes=True)),
    ('cum_m_clip', dict(units='m', cumulative=True, dtype=np.float64), dict(cumulative=True, force_positive_runoff=True)),
    ('three_files_kg', dict(units='kg m-2', files=3), dict(force_positive_runoff=True, as_volumes=True)),
    ('irregular', dict(units='m', dtype=np.float64, hours=[0, 1, 2, 4, 5, 6, 9, 10]), dict(as_volumes=True)),
    ('irregular_kept', dict(units='m', dtype=np.float64, hours=[0, 1, 2, 4, 5, 6, 9, 10]), dict(force_uniform_timesteps=False)),
]


@pytest.fixture(params=['oracle_device', pytest.param('hip', marks=pytest.mark.gpu)])
def runoff_backend(request, monkeypatch):
    """'oracle_device': host logic of river_route_amd.runoff + the oracle's arithmetic (pins both to the reference);
    'hip': the same host logic + the HIP kernel."""
    if request.param == 'oracle_device':
        from river_route_amd import engine

        def fake(indptr, indices, weights, runoff_tp, area=None, flags=0, device=0):
            W = scipy.sparse.csr_matrix((weights, indices, indptr), shape=(len(indptr) - 1, runoff_tp.shape[1]))
            return oracle.runoff_to_qlateral_core(W, runoff_tp, area, bool(flags & engine.RUNOFF_CUMULATIVE),
                                                  bool(flags & engine.RUNOFF_FORCE_POSITIVE), bool(flags & engine.RUNOFF_KEEP_NAN))
        monkeypatch.setattr(engine, 'runoff_to_qlateral', fake)
    return request.param


@pytest.mark.parametrize('case_id', range(len(GOLDEN_CASES)))
def test_against_reference_golden(tmp_path, runoff_backend, case_id):
    from river_route_amd.runoff import runoff_to_qlateral
    tag, gk, rk = GOLDEN_CASES[case_id]
    g = np.load(os.path.join(REPO, 'tests', 'golden', 'runoff.npz'))
    wfile, paths = write_grid_case(tmp_path, g[f'{tag}/table'], g[f'{tag}/grid'], g[f'{tag}/hours'], gk['units'], gk.get('files', 1))
    ds = runoff_to_qlateral(paths if len(paths) > 1 else paths[0], wfile, var_x='longitude', var_y='latitude',
                            var_t='valid_time', **rk)
    want = g[f'{tag}/qlateral']
    np.testing.assert_array_equal(ds['river_id'].values, g[f'{tag}/river_id'])
    np.testing.assert_array_equal(ds['time'].values.astype('datetime64[s]').astype(np.int64), g[f'{tag}/time'])
    assert ds['qlateral'].attrs['units'] == str(g[f'{tag}/units'])
    assert_close(ds['qlateral'].values, want, tag)


# ---- the routers' grid_runoff_files branch (TransformMuskingum.py:38-51), both backends ----
from test_routers import backend, case, drive  # noqa: E402,F401  (fixtures)
import river_route_amd as rr  # noqa: E402


@pytest.mark.parametrize('router', ['rapid', 'rapid_one_step', 'unit'])
def test_router_with_grid_runoff_files(backend, case, tmp_path, monkeypatch, router):
    """RapidMuskingum (UnitMuskingum) configured with grid_runoff_files + grid_weights_file routes exactly what it routes
    when handed the oracle's catchment volumes (depths) for the same grids."""
    from river_route_amd import engine
    from test_routers import _unit_files
    cls = rr.UnitMuskingum if router == 'unit' else rr.RapidMuskingum
    extra = dict(dt_routing=900) if router == 'rapid' else {}      # one routing step per runoff step: the fused in-pass on the GPU
    if router == 'unit':
        kp, us = _unit_files(case)
        extra = dict(dt_routing=1200, uh_kernel_file=kp, uh_state_init_file=us)
    if backend == 'oracle_injected':
        def fake(indptr, indices, weights, runoff_tp, area=None, flags=0, device=0):
            W = scipy.sparse.csr_matrix((weights, indices, indptr), shape=(len(indptr) - 1, runoff_tp.shape[1]))
            return oracle.runoff_to_qlateral_core(W, runoff_tp, area, bool(flags & 1), bool(flags & 2), bool(flags & 4))
        monkeypatch.setattr(engine, 'runoff_to_qlateral', fake)
    g = case['g']
    river_ids = g['river_ids']
    rng = np.random.default_rng(21)
    nx, ny, T = 9, 6, int(g['vol0'].shape[0])
    rows = []
    for rid in river_ids:                         # table in params order = the order the routers need
        for c in rng.choice(nx * ny, size=int(rng.integers(1, 4)), replace=False):
            rows.append((rid, c % nx, c // nx, float(rng.random()), float(rng.uniform(1e5, 1e7))))
    tab = np.array(rows)
    wfile = tmp_path / 'weights.nc'
    write_nc3(wfile, {'index': len(rows)}, {
        'river_id': (('index',), tab[:, 0].astype(np.int64), {}), 'x_index': (('index',), tab[:, 1].astype(np.int64), {}),
        'y_index': (('index',), tab[:, 2].astype(np.int64), {}), 'proportion': (('index',), tab[:, 3], {}),
        'area_sqm': (('index',), tab[:, 4], {})})
    grids, files, series = [], [], []
    for f in range(2):
        grid = (rng.random((T, ny, nx)) * 1e-3).astype(np.float32)
        secs = (case['dates'][f] - np.datetime64('1970-01-01T00:00:00')).astype('timedelta64[s]').astype(np.float64)
        p = tmp_path / f'grid{f}.nc'
        write_nc3(p, {'time': T, 'y': ny, 'x': nx}, {
            'time': (('time',), secs, {'units': 'seconds since 1970-01-01 00:00:00'}),
            'ro': (('time', 'y', 'x'), grid, {'units': 'm'})})
        files.append(str(p))
        _, vol = brute_force(tab, grid, 1.0, False, False, router != 'unit')
        series.append(vol)

    got = []
    r = cls(params_file=case['params'], grid_runoff_files=files, grid_weights_file=str(wfile),
            discharge_dir=str(tmp_path), channel_state_init_file=case['init'], log=False, **extra)
    r.set_write_discharges(lambda d, q, f_, rf='': got.append((np.asarray(d), np.asarray(q), f_, rf)))
    r.route()
    r_ref, want = drive(cls, case, series, channel_state_init_file=case['init'], **extra)
    assert len(got) == 2
    for (d, q, f_, rf), (dw, qw, _, _) in zip(got, want):
        np.testing.assert_array_equal(d, dw)
        np.testing.assert_allclose(q, qw, rtol=1.2e-7, atol=1e-10 * float(np.abs(qw).max()))
        assert rf in files and os.path.basename(f_).startswith('discharge_grid')
    np.testing.assert_allclose(r.channel_state, r_ref.channel_state, rtol=1e-10, atol=1e-10 * np.abs(r_ref.channel_state).max())

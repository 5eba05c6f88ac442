"""
Router-level parity: the host-side mirror of the reference's Router API (Configs, Muskingum, RapidMuskingum,
UnitMuskingum, UnitHydrograph, tools.adjacency_matrix) against tests/golden/routers.npz, which was produced by
running the reference's own routers on the same in-memory inputs (tests/golden/make_golden.py).

Two backends: on the GPU box the real HIP engine (`-m gpu`); on CPU the oracle is injected in place of the
engine handle so the host logic (time-step algebra, state hand-off, resampling, f32 cast, writer protocol,
config handling) is covered without a GPU.  The injection lives here, in tests/, never in the product.
"""
import json
import os

import numpy as np
import pandas as pd
import pytest
import scipy.sparse

import river_route_amd as rr
from river_route_amd import _lib
from river_route_amd.routers import muskingum as musk_mod
from river_route_amd import uhkernels as uhk_mod


class OraclePlan:
    """Stand-in for river_route_amd.engine.Plan backed by oracle/ (tests only)."""

    def __init__(self, indptr, indices, device=0):
        from conftest import unit_split
        self.indptr, self.indices = np.asarray(indptr, np.int32), np.asarray(indices, np.int32)
        self.n = len(self.indptr) - 1
        self._split = unit_split(self.indptr, self.indices, self.n)

    def close(self):
        pass

    def set_coeffs(self, lhs, c2, c3, c4_dt=None):
        self.lhs, self.c2, self.c3, self.c4 = lhs, c2, c3, c4_dt

    def rapid_route(self, q_t, ql, d, nsub):
        from oracle import oracle
        oracle.rapid_route(self.indptr, self.indices, self.lhs, self.c2, self.c3, self.c4, q_t, ql, d, nsub)

    def muskingum_route(self, q_t, d, n_out, nrpo):
        from oracle import oracle
        oracle.muskingum_route(self.indptr, self.indices, self.lhs, self.c2, self.c3, q_t, d, n_out, nrpo)

    def unit_route(self, q_ch, q_full, conv, d, nsub):
        from oracle import oracle
        hw_idx, inner_idx, A_in, A_hw = self._split
        c1 = np.zeros(self.n)
        c1[self.indices] = -self.lhs
        c1i, c2i, c3i = c1[inner_idx], self.c2[inner_idx], self.c3[inner_idx]
        oracle.unit_route(A_in.indptr, A_in.indices, -c1i[A_in.indices], A_in.indptr, A_in.indices, A_in.data,
                          A_hw.indptr, A_hw.indices, A_hw.data, c1i, c2i, c3i, hw_idx, inner_idx, q_ch, q_full,
                          np.ascontiguousarray(conv), d, nsub)


def oracle_uh_convolve(kernel, state, lateral, device=0):
    from oracle import oracle
    uh = oracle.UnitHydrograph(kernel)
    uh.state = state
    return uh.convolve(lateral)


@pytest.fixture(params=['oracle_injected', pytest.param('hip', marks=pytest.mark.gpu)])
def backend(request, monkeypatch):
    if request.param == 'oracle_injected':
        monkeypatch.setattr(musk_mod, 'Plan', OraclePlan)
        monkeypatch.setattr(uhk_mod, 'uh_convolve', oracle_uh_convolve)
    return request.param


@pytest.fixture
def case(tmp_path, golden_routers):
    g = golden_routers
    params = tmp_path / 'params.parquet'
    pd.DataFrame({'river_id': g['river_ids'], 'downstream_river_id': g['downstream_ids'], 'k': g['k'],
                  'x': g['x']}).to_parquet(params)
    init = tmp_path / 'init.parquet'
    pd.DataFrame({'Q': g['q0']}).to_parquet(init)
    files = []
    for i in range(2):
        f = tmp_path / f'ql{i}.nc'
        f.touch()
        files.append(str(f))
    dates = [g[f'dates{i}'].astype('datetime64[s]') for i in range(2)]
    return dict(g=g, tmp=tmp_path, params=str(params), init=str(init), files=files, dates=dates)


def drive(cls, case, series, **cfg):
    dates, files = case['dates'], case['files']

    class InMemory(cls):
        def _qlateral_generator(self):
            for d, a, fin, fout in zip(dates, series, self.cfg.qlateral_files, self.cfg.discharge_files):
                yield d, a, fin, fout

    got = []
    r = InMemory(params_file=case['params'], qlateral_files=files, discharge_dir=str(case['tmp']), log=False, **cfg)
    r.set_write_discharges(lambda d, q, f, rf='': got.append((np.asarray(d), np.asarray(q), f, rf)))
    assert r.route() is r
    return r, got


def check(case, tag, r, got):
    g = case['g']
    for i, (d, q, f, rf) in enumerate(got):
        assert q.dtype == np.float32
        np.testing.assert_array_equal(d.astype('datetime64[s]').astype(np.int64), g[f'{tag}/dates{i}'])
        want = g[f'{tag}/q{i}']
        # float32 outputs: <= 1 ulp(f32) of the reference
        np.testing.assert_allclose(q, want, rtol=1.2e-7, atol=1e-10 * float(np.abs(want).max()))
        assert os.path.basename(f) == f'discharge_ql{i}.nc' and rf == case['files'][i]
    want = g[f'{tag}/final_state']
    np.testing.assert_allclose(r.channel_state, want, rtol=1e-10, atol=1e-10 * float(np.abs(want).max()))


def test_rapid_sequential(backend, case):
    g = case['g']
    r, got = drive(rr.RapidMuskingum, case, [g['vol0'], g['vol1']], channel_state_init_file=case['init'],
                   dt_routing=900)
    check(case, 'rapid_seq', r, got)
    assert (r.dt_runoff, r.dt_routing, r.num_routing_steps_per_runoff) == (3600, 900, 4)


def test_rapid_resample_to_dt_discharge(backend, case):
    g = case['g']
    r, got = drive(rr.RapidMuskingum, case, [g['vol0'], g['vol1']], channel_state_init_file=case['init'],
                   dt_routing=1800, dt_discharge=3 * 3600)
    check(case, 'rapid_seq_resample', r, got)
    assert got[0][1].shape[0] == 8


def test_rapid_on_a_postorder_params_file_takes_the_direct_path(backend, case, tmp_path):
    """The golden network's table re-sorted with tools.postorder (still valid for the reference, river_route/tools.py:103-104) and the
    lateral columns with it: RapidMuskingum(config).route() gives the reference's discharge river for river, and on the GPU the
    routing calls take the direct row path -- float32 rows out fused in (rr_rapid_route_f32_dev), no record ring, no permutation pass."""
    g = case['g']
    order = rr.tools.postorder(g['river_ids'], g['downstream_ids'])
    params = tmp_path / 'params_postorder.parquet'
    pd.DataFrame({'river_id': g['river_ids'][order], 'downstream_river_id': g['downstream_ids'][order], 'k': g['k'][order], 'x': g['x'][order]}).to_parquet(params)
    init = tmp_path / 'init_postorder.parquet'
    pd.DataFrame({'Q': g['q0'][order]}).to_parquet(init)
    sorted_case = dict(case, params=str(params))
    r, got = drive(rr.RapidMuskingum, sorted_case, [g['vol0'][:, order], g['vol1'][:, order]], channel_state_init_file=str(init), runoff_processing_mode='ensemble')
    for i, (d, q, f, rf) in enumerate(got):
        want = g[f'rapid_ens/q{i}'][:, order]
        assert q.dtype == np.float32
        np.testing.assert_allclose(q, want, rtol=1.2e-7, atol=1e-10 * float(np.abs(want).max()))
    want = g['rapid_ens/final_state'][order]
    np.testing.assert_allclose(r.channel_state, want, rtol=1e-10, atol=1e-10 * float(np.abs(want).max()))
    if backend == 'hip':
        assert r._plan.direct_info()['ok'] and r._plan.last_kernel() == 'direct'


def test_rapid_ensemble_state_is_member_mean(backend, case):
    g = case['g']
    r, got = drive(rr.RapidMuskingum, case, [g['vol0'], g['vol1']], channel_state_init_file=case['init'],
                   runoff_processing_mode='ensemble')
    check(case, 'rapid_ens', r, got)


def _unit_files(case):
    g, tmp = case['g'], case['tmp']
    kp = tmp / 'uh.npz'
    scipy.sparse.save_npz(kp, scipy.sparse.csr_matrix(g['uh_kernel']))
    us = tmp / 'uhstate.parquet'
    pd.DataFrame(g['uh_state0'].T).to_parquet(us)
    return str(kp), str(us)


def test_unit_sequential_with_states(backend, case):
    g = case['g']
    kp, us = _unit_files(case)
    final_uh = case['tmp'] / 'uh_final.parquet'
    final_q = case['tmp'] / 'q_final.parquet'
    r, got = drive(rr.UnitMuskingum, case, [g['depth0'], g['depth1']], channel_state_init_file=case['init'],
                   dt_routing=1200, uh_kernel_file=kp, uh_state_init_file=us, uh_state_final_file=str(final_uh),
                   channel_state_final_file=str(final_q))
    check(case, 'unit_seq', r, got)
    scale = float(np.abs(g['unit_seq/uh_state_final']).max())
    np.testing.assert_allclose(r._uh.state, g['unit_seq/uh_state_final'], rtol=0, atol=1e-12 * scale)
    # state files: UH state is (n_basins, n_kernel_steps); channel state one column Q (io-file-schema)
    assert pd.read_parquet(final_uh).shape == (len(g['river_ids']), g['uh_kernel'].shape[0])
    np.testing.assert_array_equal(pd.read_parquet(final_q)['Q'].to_numpy(), r.channel_state)
    assert got[0][1].min() >= 0.0   # tests/test_unit_muskingum.py:60


def test_unit_ensemble_keeps_uh_state_across_members(backend, case):
    g = case['g']
    kp, _ = _unit_files(case)
    r, got = drive(rr.UnitMuskingum, case, [g['depth0'], g['depth1']], channel_state_init_file=case['init'],
                   uh_kernel_file=kp, runoff_processing_mode='ensemble')
    check(case, 'unit_ens', r, got)
    scale = float(np.abs(g['unit_ens/uh_state_final']).max())
    np.testing.assert_allclose(r._uh.state, g['unit_ens/uh_state_final'], rtol=0, atol=1e-12 * scale)


def test_muskingum_channel_only(backend, case):
    g = case['g']
    got = []
    m = rr.Muskingum(params_file=case['params'], discharge_files=[str(case['tmp'] / 'd.nc')], log=False,
                     channel_state_init_file=case['init'], dt_routing=900, dt_total=6 * 3600, dt_discharge=1800,
                     start_datetime='2021-03-04')
    m.set_write_discharges(lambda d, q, f, rf='': got.append((np.asarray(d), np.asarray(q))))
    m.route()
    np.testing.assert_array_equal(got[0][0].astype('datetime64[s]').astype(np.int64), g['musk/dates'])
    np.testing.assert_allclose(got[0][1], g['musk/q'], rtol=1.2e-7, atol=1e-10 * float(np.abs(g['musk/q']).max()))
    np.testing.assert_allclose(m.channel_state, g['musk/final_state'], rtol=1e-10)
    assert got[0][1].shape == (12, len(g['river_ids'])) and got[0][1].min() >= 0


def test_second_route_continues_from_state(backend, case):
    """Muskingum.py:117-118: channel_state persists on the object, so route() twice == one longer run."""
    g = case['g']
    r, got = drive(rr.RapidMuskingum, case, [g['vol0'], g['vol1']], channel_state_init_file=case['init'],
                   dt_routing=900)
    first = r.channel_state.copy()
    r.route()
    assert not np.allclose(first, r.channel_state)


def test_netcdf_files_end_to_end(backend, case):
    """Real files both sides: qlateral netCDF in, discharge netCDF out (NetCDF-3 via scipy when netCDF4 is absent)."""
    from scipy.io import netcdf_file
    from river_route_amd.io import read_qlateral
    g, tmp = case['g'], case['tmp']
    for i in range(2):
        with netcdf_file(case['files'][i], 'w', version=2) as ds:
            T, n = g[f'vol{i}'].shape
            ds.createDimension('time', T)
            ds.createDimension('river_id', n)
            tv = ds.createVariable('time', 'f8', ('time',))
            tv.units = 'seconds since 1970-01-01 00:00:00'
            tv[:] = g[f'dates{i}'].astype(np.float64)
            v = ds.createVariable('qlateral', 'f8', ('time', 'river_id'))
            v[:] = g[f'vol{i}']
    cfg = tmp / 'config.yaml'
    import yaml
    cfg.write_text(yaml.safe_dump(dict(params_file=case['params'], qlateral_files=case['files'],
                                       discharge_dir=str(tmp), channel_state_init_file=case['init'],
                                       dt_routing=900, log=False)))
    r = rr.RapidMuskingum(str(cfg)).route()
    for i in range(2):
        out = tmp / f'discharge_ql{i}.nc'
        assert out.exists()
        with netcdf_file(str(out), 'r', mmap=False) as ds:
            q = np.array(ds.variables['Q'][:])
            ids = np.array(ds.variables['river_id'][:])
            units = ds.variables['time'].units
        assert q.dtype.kind == 'f' and q.dtype.itemsize == 4 and (units.decode() if isinstance(units, bytes) else units).startswith('seconds since')
        np.testing.assert_array_equal(ids, g['river_ids'])
        want = g[f'rapid_seq/q{i}']
        np.testing.assert_allclose(q, want, rtol=1.2e-7, atol=1e-10 * float(np.abs(want).max()))
    jcfg = tmp / 'config.json'
    jcfg.write_text(json.dumps(dict(params_file=case['params'], qlateral_files=case['files'],
                                    discharge_dir=str(tmp), log=False)))
    assert rr.RapidMuskingum(str(jcfg), dt_routing=900).cfg.dt_routing == 900   # kwargs override the file


def test_float32_qlateral_file_routes_like_its_float64_copy(backend, case):
    """A qlateral file that stores float32 (half the bytes) is uploaded as float32 and converted on its way into the engine's
    records (rr_rapid_route_f32in_dev); the discharge equals that of the same values stored as float64, bit for bit."""
    from scipy.io import netcdf_file
    g, tmp = case['g'], case['tmp']
    results = {}
    for kind in ('f4', 'f8'):
        files = []
        for i in range(2):
            path = str(tmp / f'{kind}_ql{i}.nc')
            files.append(path)
            with netcdf_file(path, 'w', version=2) as ds:
                T, n = g[f'vol{i}'].shape
                ds.createDimension('time', T)
                ds.createDimension('river_id', n)
                tv = ds.createVariable('time', 'f8', ('time',))
                tv.units = 'seconds since 1970-01-01 00:00:00'
                tv[:] = g[f'dates{i}'].astype(np.float64)
                v = ds.createVariable('qlateral', kind, ('time', 'river_id'))
                v[:] = g[f'vol{i}'].astype(np.float32)      # the same float32-representable values in both files
        out = []
        (tmp / kind).mkdir()
        r = rr.RapidMuskingum(params_file=case['params'], qlateral_files=files, discharge_dir=str(tmp / kind), channel_state_init_file=case['init'],
                              dt_routing=900, log=False)
        r.set_write_discharges(lambda dates, q, q_file, routed_file='': out.append(np.array(q)))
        r.route()
        results[kind] = (out, r.channel_state.copy())
    for a, b in zip(results['f4'][0], results['f8'][0]):
        assert a.dtype == np.float32
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(results['f4'][1], results['f8'][1])


@pytest.mark.gpu
@pytest.mark.parametrize('record', [False, True])
def test_float32_file_goes_from_file_to_file_without_a_host_array(case, record, monkeypatch):
    """A float32 NetCDF-3 qlateral file (fixed variable, or `time` as the record dimension: rows with the time values between them) is
    routed file to file: nc3.locate_rows -> engine.rows_upload -> rr_rapid_route_f32in_dev with the big-endian bytes converted in the
    kernels -> nc3.create_discharge_file -> engine.rows_download; io.read_qlateral and io.write_discharge are never called.  The
    discharge file read back through scipy equals what the same files give through the array path (a custom writer switches it on),
    bit for bit; layout and attributes are the reference's (Muskingum.py:337-351)."""
    from scipy.io import netcdf_file
    from river_route_amd import io as rr_io
    g, tmp = case['g'], case['tmp']
    files = []
    series = [np.vstack([g['vol0'], g['vol1']]), np.vstack([g['vol1'], g['vol0']])]      # 48 hourly rows a file: the time-tiled kernel takes calls of 32 rows or more
    for i in range(2):
        path = str(tmp / f'f4_{int(record)}_ql{i}.nc')
        files.append(path)
        with netcdf_file(path, 'w', version=2) as ds:
            T, n = series[i].shape
            ds.createDimension('time', None if record else T)
            ds.createDimension('river_id', n)
            tv = ds.createVariable('time', 'f8', ('time',))
            tv.units = 'seconds since 1970-01-01 00:00:00'
            tv[:] = float(g['dates0'][0]) + 3600.0 * (T * i + np.arange(T))
            v = ds.createVariable('qlateral', 'f4', ('time', 'river_id'))
            v[:] = series[i].astype(np.float32)
    kw = dict(params_file=case['params'], qlateral_files=files, channel_state_init_file=case['init'], dt_routing=3600, dt_discharge=2 * 3600, log=False)
    (tmp / 'arrays').mkdir()
    want = []
    r0 = rr.RapidMuskingum(discharge_dir=str(tmp / 'arrays'), **kw)
    r0.set_write_discharges(lambda dates, q, q_file, routed_file='': want.append((np.array(dates), np.array(q))))      # a custom writer: the array path
    r0.route()
    (tmp / 'files').mkdir()

    def never(*a, **k):
        raise AssertionError('the file-to-file path must not read or write the block as a host array')
    monkeypatch.setattr(rr_io, 'read_qlateral', never)
    monkeypatch.setattr(rr_io, 'write_discharge', never)
    r1 = rr.RapidMuskingum(discharge_dir=str(tmp / 'files'), **kw).route()
    np.testing.assert_array_equal(r1.channel_state, r0.channel_state)
    for i in range(2):
        with netcdf_file(str(tmp / 'files' / f'discharge_f4_{int(record)}_ql{i}.nc'), 'r', mmap=False) as ds:
            q = np.array(ds.variables['Q'][:], dtype=np.float32)
            np.testing.assert_array_equal(q, want[i][1])
            np.testing.assert_array_equal(np.array(ds.variables['river_id'][:]), g['river_ids'])
            secs = np.array(ds.variables['time'][:])
            assert ds.variables['time'].units.decode() == 'seconds since ' + str(want[i][0][0].astype('datetime64[s]')).replace('T', ' ')
            np.testing.assert_array_equal(secs, (want[i][0] - want[i][0][0]).astype('timedelta64[s]').astype(np.float64))
            assert ds.variables['Q'].units == b'm3 s-1' and ds.runoff_file.decode() == files[i]
    # three rows per output row do not divide a batch of 128: the fused float32 form does not apply, and the file is routed as an array
    monkeypatch.undo()
    (tmp / 'thirds').mkdir()
    r2 = rr.RapidMuskingum(discharge_dir=str(tmp / 'thirds'), **dict(kw, dt_discharge=3 * 3600)).route()
    with netcdf_file(str(tmp / 'thirds' / f'discharge_f4_{int(record)}_ql0.nc'), 'r', mmap=False) as ds:
        assert ds.variables['Q'].shape[0] == series[0].shape[0] // 3
    np.testing.assert_array_equal(r2.channel_state, r0.channel_state)


@pytest.mark.gpu
def test_float32_depth_file_goes_from_file_to_file_unit_muskingum(case, monkeypatch):
    """UnitMuskingum on float32 runoff-depth files with `time` as the record dimension: file to file (rows_upload, the convolution fused
    into the routing call with the big-endian rows converted in the kernels -- the pass that builds the records and the one that carries the
    convolution's tail --, rows_download), against the array path on the same files: discharge, router state and UH state bit for bit."""
    from scipy.io import netcdf_file
    from river_route_amd import io as rr_io
    g, tmp = case['g'], case['tmp']
    kp, us = _unit_files(case)
    series = [np.vstack([g['depth0'], g['depth1']]), np.vstack([g['depth1'], g['depth0']])]
    files = []
    for i in range(2):
        path = str(tmp / f'depth_f4_{i}.nc')
        files.append(path)
        with netcdf_file(path, 'w', version=2) as ds:
            T, n = series[i].shape
            ds.createDimension('time', None)
            ds.createDimension('river_id', n)
            tv = ds.createVariable('time', 'f8', ('time',))
            tv.units = 'seconds since 1970-01-01 00:00:00'
            tv[:] = float(g['dates0'][0]) + 3600.0 * (T * i + np.arange(T))
            v = ds.createVariable('qlateral', 'f4', ('time', 'river_id'))
            v[:] = series[i].astype(np.float32)
    kw = dict(params_file=case['params'], qlateral_files=files, channel_state_init_file=case['init'], uh_kernel_file=kp, uh_state_init_file=us, log=False)
    (tmp / 'arrays').mkdir()
    want = []
    r0 = rr.UnitMuskingum(discharge_dir=str(tmp / 'arrays'), **kw)
    r0.set_write_discharges(lambda dates, q, q_file, routed_file='': want.append(np.array(q)))
    r0.route()
    (tmp / 'files').mkdir()

    def never(*a, **k):
        raise AssertionError('the file-to-file path must not read or write the block as a host array')
    monkeypatch.setattr(rr_io, 'read_qlateral', never)
    monkeypatch.setattr(rr_io, 'write_discharge', never)
    r1 = rr.UnitMuskingum(discharge_dir=str(tmp / 'files'), **kw).route()
    np.testing.assert_array_equal(r1.channel_state, r0.channel_state)
    np.testing.assert_array_equal(r1._uh.state, r0._uh.state)
    for i in range(2):
        with netcdf_file(str(tmp / 'files' / f'discharge_depth_f4_{i}.nc'), 'r', mmap=False) as ds:
            np.testing.assert_array_equal(np.array(ds.variables['Q'][:], dtype=np.float32), want[i])


def test_float32_runoff_depths_route_like_their_float64_copy(backend, case):
    """UnitMuskingum fed float32 runoff depths (what a float32 file yields) takes them to the device as float32
    (rr_unit_route_uh_f32in_dev where the engine offers it, its float64 path otherwise): discharge, router state and UH state are
    those of the same values as float64, bit for bit."""
    g = case['g']
    kp, us = _unit_files(case)
    d32 = [g['depth0'].astype(np.float32), g['depth1'].astype(np.float32)]
    runs = []
    for series in (d32, [d.astype(np.float64) for d in d32]):
        r, got = drive(rr.UnitMuskingum, case, series, channel_state_init_file=case['init'], dt_routing=1200, uh_kernel_file=kp,
                       uh_state_init_file=us)
        runs.append((r, got))
    for (_, q32, _, _), (_, q64, _, _) in zip(runs[0][1], runs[1][1]):
        assert q32.dtype == np.float32
        np.testing.assert_array_equal(q32, q64)
    np.testing.assert_array_equal(runs[0][0].channel_state, runs[1][0].channel_state)
    np.testing.assert_array_equal(runs[0][0]._uh.state, runs[1][0]._uh.state)


# ---------------------------------------------------------------- config / validation behaviour (no compute)

def test_configs_validation(tmp_path):
    params = tmp_path / 'p.parquet'
    params.touch()
    with pytest.raises(ValueError, match='Provide discharge_dir'):
        rr.Configs(params_file=str(params))
    with pytest.raises(ValueError, match='not both'):
        rr.Configs(params_file=str(params), discharge_dir=str(tmp_path), discharge_files=['a.nc'])
    with pytest.raises(ValueError, match='Missing required config: params_file'):
        rr.Configs(discharge_dir=str(tmp_path))
    with pytest.raises(FileNotFoundError, match='params_file not found'):
        rr.Configs(params_file=str(tmp_path / 'nope.parquet'), discharge_dir=str(tmp_path))
    with pytest.raises(FileNotFoundError, match='qlateral_files: .* not found'):
        rr.Configs(params_file=str(params), discharge_dir=str(tmp_path), qlateral_files=['missing.nc'])
    with pytest.raises(NotADirectoryError, match='Output directory not found'):
        rr.Configs(params_file=str(params), discharge_dir=str(tmp_path / 'nodir'))
    with pytest.raises(NotADirectoryError, match='specified output path'):
        rr.Configs(params_file=str(params), discharge_files=[str(tmp_path / 'nodir' / 'd.nc')])
    with pytest.raises(ValueError, match="runoff_processing_mode must be one of"):
        rr.Configs(params_file=str(params), discharge_dir=str(tmp_path), runoff_processing_mode='parallel')
    with pytest.raises(TypeError):
        rr.Configs(params_file=str(params), discharge_dir=str(tmp_path), not_a_key=1)
    ql = tmp_path / 'jan.nc'
    ql.touch()
    c = rr.Configs(params_file='p.parquet' if False else str(params), discharge_dir=str(tmp_path), qlateral_files=str(ql),
                   log=False)
    assert c.qlateral_files == [str(ql)] and c.discharge_files == [str(tmp_path / 'discharge_jan.nc')]
    assert c.progress_bar is False and os.path.isabs(c.params_file)
    assert rr.Configs(params_file=str(params), discharge_dir=str(tmp_path)).discharge_files == \
        [str(tmp_path / 'discharge.nc')]


def test_router_required_keys_and_bad_config_suffix(tmp_path):
    params = tmp_path / 'p.parquet'
    params.touch()
    with pytest.raises(RuntimeError, match='Unrecognized simulation config file type'):
        rr.Muskingum(str(tmp_path / 'config.toml'))
    with pytest.raises(ValueError, match='channel_state_init_file is required for Muskingum'):
        rr.Muskingum(params_file=str(params), discharge_dir=str(tmp_path), log=False).route()
    with pytest.raises(ValueError, match='uh_kernel_file is required for UnitMuskingum'):
        rr.UnitMuskingum(params_file=str(params), discharge_dir=str(tmp_path), log=False).route()
    with pytest.raises(ValueError, match='Provide qlateral_files or grid_runoff_files'):
        rr.RapidMuskingum(params_file=str(params), discharge_dir=str(tmp_path), log=False).route()


def test_time_step_rules(backend, case):
    g = case['g']
    with pytest.raises(ValueError, match='dt_runoff must be an integer multiple of dt_routing'):
        drive(rr.RapidMuskingum, case, [g['vol0'], g['vol1']], dt_routing=1000)
    with pytest.raises(ValueError, match='dt_runoff must be >= dt_routing'):
        drive(rr.RapidMuskingum, case, [g['vol0'], g['vol1']], dt_routing=7200)
    with pytest.raises(ValueError, match='dt_total must be an integer multiple of dt_discharge'):
        drive(rr.RapidMuskingum, case, [g['vol0'], g['vol1']], dt_discharge=5 * 3600)


def test_adjacency_matrix_matches_reference_structure(golden_kernels):
    g = golden_kernels
    for tag in ('docs9', 'tree1k', 'forest30'):
        A = rr.tools.adjacency_matrix(g[f'{tag}/river_ids'], g[f'{tag}/downstream_ids'])
        assert isinstance(A, scipy.sparse.csc_matrix) and A.dtype == np.float64
        np.testing.assert_array_equal(A.indptr, g[f'{tag}/indptr'])
        np.testing.assert_array_equal(A.indices, g[f'{tag}/indices'])
    with pytest.raises(ValueError, match='topologically sorted'):          # tests/test_tools.py:48-53
        rr.tools.adjacency_matrix(np.array([10, 20, 30]), np.array([20, -1, 10]))
    with pytest.raises(ValueError, match='Unknown downstream_river_id'):   # tests/test_tools.py:56-60
        rr.tools.adjacency_matrix(np.array([10, 20]), np.array([-1, 999]))


def test_engine_order_is_a_valid_topological_order():
    net = rr.synth.synth_network(3000)
    order = rr.tools.engine_order(net.river_ids, net.downstream_ids)
    rid, did = net.river_ids[order], net.downstream_ids[order]
    A = rr.tools.adjacency_matrix(rid, did)      # raises if not upstream -> downstream
    assert A.nnz == net.n - 1
    from river_route_amd.engine import Plan
    with Plan(A.indptr, A.indices, device=_lib.RR_DEVICE_NONE) as plan:
        assert plan.identity_order


@pytest.mark.gpu
def test_device_file_path_equals_host_path(case, monkeypatch):
    """The routers keep each file on the GPU (lateral in, resample-mean + float32 cast on the device, float32 out);
    the reference-shaped host path (_router + numpy post-processing) must give the same arrays."""
    from river_route_amd.routers.transform import TransformMuskingum
    g = case['g']
    kp, us = _unit_files(case)
    runs = {}
    for flag in (True, False):
        monkeypatch.setattr(TransformMuskingum, '_device_postprocess', flag)
        r1, got1 = drive(rr.RapidMuskingum, case, [g['vol0'], g['vol1']], channel_state_init_file=case['init'],
                         dt_routing=1800, dt_discharge=3 * 3600)
        r2, got2 = drive(rr.UnitMuskingum, case, [g['depth0'], g['depth1']], channel_state_init_file=case['init'],
                         dt_routing=1200, uh_kernel_file=kp, uh_state_init_file=us, dt_discharge=2 * 3600)
        runs[flag] = (r1.channel_state, [q for _, q, _, _ in got1], r2.channel_state, [q for _, q, _, _ in got2],
                      r2._uh.state)
    for a, b in zip(runs[True], runs[False]):
        if isinstance(a, list):
            for x, y in zip(a, b):
                assert x.dtype == np.float32 and x.shape == y.shape
                np.testing.assert_array_equal(x, y)
        else:
            np.testing.assert_array_equal(a, b)


# SURVEY section 8 row f3 on both boxes: the builders are host code (pinned by reference goldens, runs here); on the GPU
# box (`-m gpu`) the same kernels also drive the HIP convolution against the oracle.
BOX = pytest.mark.parametrize('box', ['host', pytest.param('gpu_box', marks=pytest.mark.gpu)])


@BOX
def test_scs_kernel_builders(golden_kernels, box):
    """uhkernels/_SCSBase.py: kernels equal the reference's, conserve volume (tests/test_uhkernels.py:19-42)."""
    from river_route_amd.uhkernels import SCSCurvilinear, SCSTriangular
    g = golden_kernels
    tc, area = g['scs/tc'], g['scs/area']
    for tr in (3600.0, 900.0):
        for name, cls in (('triangular', SCSTriangular), ('curvilinear', SCSCurvilinear)):
            uh = cls(tr=tr, tc=tc, area=area)
            want = g[f'scs/tr{int(tr)}/{name}']
            assert uh.kernel.shape == want.shape and uh.kernel.ndim == 2
            np.testing.assert_allclose(uh.kernel, want, rtol=1e-12, atol=1e-12 * want.max())
            np.testing.assert_allclose(uh.kernel.sum(axis=0) * tr, area, rtol=1e-6)
            if box == 'gpu_box':      # UnitHydrograph.convolve (UnitHydrograph.py:77-107) with this kernel: HIP engine vs oracle, two files
                from oracle import oracle
                from river_route_amd.engine import uh_convolve
                ref, state = oracle.UnitHydrograph(uh.kernel), np.zeros_like(uh.kernel)
                for f, T in enumerate((70, 5)):
                    depth = 1e-3 * np.random.default_rng(f).random((T, area.shape[0]))
                    want_rows = ref.convolve(depth)
                    got_rows = uh_convolve(uh.kernel, state, depth)
                    np.testing.assert_allclose(got_rows, want_rows, rtol=1e-10, atol=1e-10 * np.abs(want_rows).max())
                    np.testing.assert_allclose(state, ref.state, rtol=1e-10, atol=1e-10 * np.abs(want_rows).max())
    with pytest.raises(ValueError, match='tr must be > 0'):
        SCSTriangular(tr=0.0, tc=tc, area=area)
    with pytest.raises(ValueError, match='same length'):
        SCSTriangular(tr=900.0, tc=tc, area=area[:2])


@BOX
def test_scs_kernel_file_round_trip(tmp_path, box):
    from river_route_amd.uhkernels import SCSTriangular, UnitHydrograph
    uh = SCSTriangular(tr=900.0, tc=np.array([2000.0, 9000.0]), area=np.array([1e6, 4e6]))
    path = tmp_path / 'k.npz'
    uh.save(path)
    loaded = scipy.sparse.load_npz(path).toarray()
    np.testing.assert_array_equal(loaded, uh.kernel)


def test_deep_validate(tmp_path, golden_routers):
    """Config.py:185-280 -- explicit content validation with the reference's messages."""
    g = golden_routers
    good = tmp_path / 'p.parquet'
    df = pd.DataFrame({'river_id': g['river_ids'], 'downstream_river_id': g['downstream_ids'], 'k': g['k'], 'x': g['x']})
    df.to_parquet(good)
    st = tmp_path / 's.parquet'
    pd.DataFrame({'Q': g['q0']}).to_parquet(st)
    cfg = rr.Configs(params_file=str(good), discharge_dir=str(tmp_path), channel_state_init_file=str(st))
    assert cfg.deep_validate() is cfg

    def broken(mutate, match, state=None):
        bad = tmp_path / 'bad.parquet'
        d = df.copy()
        mutate(d)
        d.to_parquet(bad)
        c = rr.Configs(params_file=str(bad), discharge_dir=str(tmp_path), channel_state_init_file=state)
        with pytest.raises(ValueError, match=match):
            c.deep_validate()

    broken(lambda d: d.drop(columns=['k'], inplace=True), 'missing k column')
    broken(lambda d: d.__setitem__('k', -d['k']), 'k column must be positive')
    broken(lambda d: d.__setitem__('x', d['x'] + 1.0), r'x column must be in the range \[0, 0.5\]')
    broken(lambda d: d.__setitem__('river_id', d['river_id'].iloc[0]), 'river_id column must be unique')
    broken(lambda d: d.__setitem__('downstream_river_id', d['downstream_river_id'].where(d.index != 3, 99999999)),
           'must exist in river_id')
    broken(lambda d: d.__setitem__('downstream_river_id', d['downstream_river_id'].astype(float)), 'must be integer type')

    def reverse(d):
        d.iloc[:] = d.iloc[::-1].to_numpy()
        for c in ('river_id', 'downstream_river_id'):
            d[c] = d[c].astype(np.int64)
    broken(reverse, 'not topologically sorted')
    short = tmp_path / 'short.parquet'
    pd.DataFrame({'Q': g['q0'][:-1]}).to_parquet(short)
    broken(lambda d: None, 'same number of rows', state=str(short))
    neg = tmp_path / 'neg.parquet'
    pd.DataFrame({'Q': -g['q0']}).to_parquet(neg)
    broken(lambda d: None, 'non-negative', state=str(neg))


def test_cli_dispatch(backend, case, capsys):
    """river_route/_cli.py: `rr RapidMuskingum cfg` and `rr route --router X cfg` run Router(cfg).route()."""
    import yaml
    from scipy.io import netcdf_file
    from river_route_amd import _cli
    g, tmp = case['g'], case['tmp']
    for i in range(2):
        with netcdf_file(case['files'][i], 'w', version=2) as ds:
            T, n = g[f'vol{i}'].shape
            ds.createDimension('time', T)
            ds.createDimension('river_id', n)
            tv = ds.createVariable('time', 'f8', ('time',))
            tv.units = 'seconds since 1970-01-01 00:00:00'
            tv[:] = g[f'dates{i}'].astype(np.float64)
            ds.createVariable('qlateral', 'f8', ('time', 'river_id'))[:] = g[f'vol{i}']
    cfg = tmp / 'c.yaml'
    cfg.write_text(yaml.safe_dump(dict(params_file=case['params'], qlateral_files=case['files'],
                                       discharge_dir=str(tmp), log=False, progress_bar=False)))
    _cli.main(['RapidMuskingum', str(cfg)])
    assert (tmp / 'discharge_ql0.nc').exists() and (tmp / 'discharge_ql1.nc').exists()
    (tmp / 'discharge_ql0.nc').unlink()
    _cli.main(['route', '--router', 'RapidMuskingum', str(cfg)])
    assert (tmp / 'discharge_ql0.nc').exists()
    _cli.main([])
    assert 'usage' in capsys.readouterr().out.lower()

"""river_route_amd.nc3: where a NetCDF-3 variable's rows lie (against files scipy writes) and the discharge file written without its rows
(read back by scipy: the reference's layout, river_route/routers/Muskingum.py:337-351)."""
import numpy as np
import pytest
from scipy.io import netcdf_file

from river_route_amd import nc3


def _write(path, data, record, extra_record_var=True, version=2):
    T, n = data.shape
    with netcdf_file(str(path), 'w', version=version) as ds:
        ds.createDimension('time', None if record else T)
        ds.createDimension('river_id', n)
        if extra_record_var or not record:
            tv = ds.createVariable('time', 'f8', ('time',))
            tv.units = 'seconds since 2020-01-01 00:00:00'
            tv[:] = np.arange(T) * 3600.0
        rid = ds.createVariable('river_id', 'i4', ('river_id',))
        rid[:] = 7 + np.arange(n)
        v = ds.createVariable('qlateral', data.dtype.str[1:], ('time', 'river_id'))
        v[:] = data


@pytest.mark.parametrize('record,extra,version,dtype,n', [(False, True, 2, 'f4', 37), (True, True, 2, 'f4', 37), (True, False, 2, 'f4', 5), (True, True, 1, 'f8', 11),
                                                           (False, True, 1, 'f4', 1), (True, True, 2, 'f4', 3)])
def test_locate_rows_in_files_scipy_writes(tmp_path, record, extra, version, dtype, n):
    T = 9
    data = np.random.default_rng(3).random((T, n)).astype(dtype)
    path = tmp_path / 'q.nc'
    _write(path, data, record, extra, version)
    blk = nc3.locate_rows(path, 'qlateral')
    assert blk is not None and (blk.rows, blk.cols) == (T, n) and blk.big_endian and blk.dtype.itemsize == np.dtype(dtype).itemsize
    raw = np.fromfile(path, dtype=np.uint8)
    for r in range(T):
        row = raw[blk.offset + r * blk.pitch: blk.offset + r * blk.pitch + blk.row_bytes].view(blk.dtype)
        np.testing.assert_array_equal(row.astype(dtype), data[r])
    if extra or not record:
        t, atts = nc3.read_vector(path, 'time')
        np.testing.assert_array_equal(t, np.arange(T) * 3600.0)
        assert atts['units'] == 'seconds since 2020-01-01 00:00:00'
    ids, _ = nc3.read_vector(path, 'river_id')
    np.testing.assert_array_equal(ids, 7 + np.arange(n))
    assert nc3.locate_rows(path, 'nothing') is None and nc3.locate_rows(path, 'river_id') is None


def test_other_formats_are_left_to_their_library(tmp_path):
    p = tmp_path / 'x.nc'
    p.write_bytes(b'\x89HDF\r\n\x1a\n' + b'\x00' * 64)
    assert nc3.locate_rows(p, 'qlateral') is None
    p.write_bytes(b'CD')
    assert nc3.locate_rows(p, 'qlateral') is None


@pytest.mark.parametrize('record', [False, True])
def test_discharge_file_without_its_rows_reads_back_through_scipy(tmp_path, record):
    T, n = 6, 23
    dates = np.datetime64('2021-03-04T00:00:00', 's') + np.arange(T) * np.timedelta64(3 * 3600, 's')
    ids = 1000 + 3 * np.arange(n)
    q = np.random.default_rng(5).random((T, n)).astype(np.float32)
    path = tmp_path / 'discharge.nc'
    blk = nc3.create_discharge_file(path, dates, ids, routed_file='/some/where/ql.nc', record_dim=record)
    assert (blk.rows, blk.cols, blk.dtype) == (T, n, np.dtype('>f4'))
    mm = np.memmap(str(path), dtype=np.uint8, mode='r+')      # what engine.rows_download does: row r at offset + r * pitch
    for r in range(T):
        mm[blk.offset + r * blk.pitch: blk.offset + r * blk.pitch + blk.row_bytes] = q[r].astype('>f4').view(np.uint8)
    mm.flush()
    del mm
    with netcdf_file(str(path), 'r', mmap=False) as ds:
        assert set(ds.variables) == {'time', 'river_id', 'Q'}
        assert ds.variables['Q'].dimensions == ('time', 'river_id') and ds.variables['Q'].shape == (T, n)
        np.testing.assert_array_equal(np.array(ds.variables['Q'][:], dtype=np.float32), q)
        np.testing.assert_array_equal(np.array(ds.variables['river_id'][:]), ids)
        np.testing.assert_array_equal(np.array(ds.variables['time'][:]), np.arange(T) * 3 * 3600.0)
        assert ds.variables['time'].units == b'seconds since 2021-03-04 00:00:00'
        assert ds.variables['Q'].units == b'm3 s-1' and ds.variables['Q'].aggregation_method == b'mean' and ds.variables['Q'].standard_name == b'discharge'
        assert ds.runoff_file == b'/some/where/ql.nc'
        assert (ds.dimensions['time'] is None) == record

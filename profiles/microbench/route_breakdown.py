"""Where RapidMuskingum(config).route() spends its wall time at 1M reaches x 744 hourly rows (float32 qlateral netCDF in, float32 discharge
netCDF out; the shape of bench.py's `bench_dropin`): every stage wrapped by a timer, peak host RSS.  usage: route_breakdown.py [n] [T] [order]"""
import os
import resource
import sys
import tempfile
import time
from collections import OrderedDict

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import pandas as pd  # noqa: E402
from scipy.io import netcdf_file  # noqa: E402

import river_route_amd as rr  # noqa: E402
from river_route_amd import engine, io, nc3, synth, tools  # noqa: E402
from river_route_amd.routers import _device, muskingum  # noqa: E402

try:
    import pyarrow  # noqa: F401
except ImportError:
    pd.read_parquet = lambda path, columns=None, **kw: (pd.read_pickle(path)[list(columns)] if columns is not None else pd.read_pickle(path))
    pd.DataFrame.to_parquet = lambda self, path, **kw: self.to_pickle(path)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 744
order = sys.argv[3] if len(sys.argv) > 3 else 'random'
dt = 3600.0
times = OrderedDict()


def timed(name, fn):
    def wrapper(*a, **kw):
        t0 = time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            times[name] = times.get(name, 0.0) + time.perf_counter() - t0
    return wrapper


tmp = os.environ.get('RR_BREAKDOWN_DIR') or tempfile.mkdtemp(prefix='rr_breakdown_')
params = os.path.join(tmp, 'params.parquet')
qfile = os.path.join(tmp, 'qlateral.nc')
if os.environ.get('RR_BREAKDOWN_MAKE'):      # the child that writes the input files: its arrays must not count in the router's peak RSS
    net = synth.synth_network(n, order=order)
    pd.DataFrame({'river_id': net.river_ids, 'downstream_river_id': net.downstream_ids, 'k': net.k, 'x': net.x}).to_parquet(params)
    ql32 = synth.synth_qlateral(n, 0, T, dt=dt).astype(np.float32)
    dates = (np.datetime64('2020-01-01T00:00:00', 's') + np.arange(T) * np.timedelta64(int(dt), 's')).astype('datetime64[s]').astype(np.int64).astype(np.float64)
    with netcdf_file(qfile, 'w', version=2) as ds:
        ds.createDimension('time', None)
        ds.createDimension('river_id', n)
        tv = ds.createVariable('time', 'f8', ('time',))
        tv.units = 'seconds since 1970-01-01 00:00:00'
        tv[:] = dates
        rid = ds.createVariable('river_id', 'i4', ('river_id',))
        rid[:] = net.river_ids.astype(np.int32)
        v = ds.createVariable('qlateral', 'f4', ('time', 'river_id'))
        v[:] = ql32
    os.makedirs(os.path.join(tmp, 'out'), exist_ok=True)
    sys.exit(0)
import subprocess
subprocess.check_call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(os.environ, RR_BREAKDOWN_MAKE='1', RR_BREAKDOWN_DIR=tmp))
print(f'input files written by a child process; this process before the first route(): peak host RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.2f} GB')

pd.read_parquet = timed('params table read (pd.read_parquet)', pd.read_parquet)
muskingum.adjacency_matrix = timed('tools.adjacency_matrix', muskingum.adjacency_matrix)
muskingum.Plan = timed('Plan(): network analysis, tile + direct plans, device upload', muskingum.Plan)
io.read_qlateral = timed('io.read_qlateral (netCDF read, byte order)', io.read_qlateral)
io.write_discharge = timed('io.write_discharge (netCDF write, byte order)', io.write_discharge)
_device.Arena.put = timed('Arena.put (hipMalloc + pageable hipMemcpy up)', _device.Arena.put)
engine.DeviceBuffer.download = timed('DeviceBuffer.download (pageable hipMemcpy down)', engine.DeviceBuffer.download)
engine.Plan.set_coeffs = timed('Plan.set_coeffs', engine.Plan.set_coeffs)
engine.rows_upload = timed('engine.rows_upload (file -> pinned chunks -> device)', engine.rows_upload)
engine.rows_download = timed('engine.rows_download (device -> pinned chunks -> file)', engine.rows_download)
nc3.create_discharge_file = timed('nc3.create_discharge_file (header, time, ids)', nc3.create_discharge_file)
nc3.locate_rows = timed('nc3.locate_rows + read_vector (header)', nc3.locate_rows)
for name in ('rapid_route_f32in_dev', 'rapid_route_f32_dev', 'rapid_route_dev', 'reserve'):
    setattr(engine.Plan, name, timed(f'Plan.{name} (enqueue)', getattr(engine.Plan, name)))

for rep in range(2):
    times.clear()
    t0 = time.perf_counter()
    r = rr.RapidMuskingum(params_file=params, qlateral_files=[qfile], discharge_dir=os.path.join(tmp, 'out'), dt_routing=int(dt), log=False)
    r.route()
    wall = time.perf_counter() - t0
    kern = r._plan.last_kernel()
    print(f'--- pass {rep}: RapidMuskingum(config).route(), {n} reaches x {T} rows, params order {order}, routing kernel {kern}: {wall:.3f} s wall, '
          f'{n * T / wall:.3e} reach-steps/s, peak host RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.1f} GB')
    acc = 0.0
    for k, v in times.items():
        print(f'    {v * 1e3:9.1f} ms  {k}')
        acc += v
    print(f'    {(wall - acc) * 1e3:9.1f} ms  everything else (config, coefficients, state tables, python)')
import shutil
shutil.rmtree(tmp, ignore_errors=True)

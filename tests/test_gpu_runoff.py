"""HIP kernel of SURVEY section 8 row f2 (rr_runoff_to_qlateral, k_runoff_to_qlateral) against the oracle restatement
of river_route/runoff.py:296-330, through the C ABI: both runoff layouts, float32 / float64 grids, every flag
combination, NaNs, empty weight rows, time lengths that are not a multiple of the kernel's 16-row chunk."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import REPO, assert_close
from test_runoff import random_weights

from oracle import oracle  # noqa: E402

pytestmark = pytest.mark.gpu


def case(n_rivers, n_points, T, dtype, cumulative, seed=0):
    rng = np.random.default_rng(seed)
    W = random_weights(rng, n_rivers, n_points)
    runoff = (rng.random((T, n_points)) * 2.0 - 0.3).astype(dtype)
    if T > 4:
        runoff[rng.integers(0, T, 7), rng.integers(0, n_points, 7)] = np.nan
    if cumulative:
        runoff = np.cumsum(np.nan_to_num(runoff), axis=0).astype(dtype)
        runoff[T // 2, min(3, n_points - 1)] = np.nan
    area = rng.uniform(1e5, 5e7, n_rivers)
    return W, runoff, area


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('cumulative,clip,volumes,keep_nan', [(False, False, False, False), (True, False, True, False),
                                                              (False, True, True, False), (True, True, False, True)])
@pytest.mark.parametrize('n_rivers,n_points,T', [(3000, 700, 37), (257, 40, 16), (1, 5, 1)])
def test_kernel_vs_oracle(n_rivers, n_points, T, dtype, cumulative, clip, volumes, keep_nan):
    from river_route_amd import engine
    W, runoff, area = case(n_rivers, n_points, T, dtype, cumulative)
    flags = (engine.RUNOFF_CUMULATIVE if cumulative else 0) | (engine.RUNOFF_FORCE_POSITIVE if clip else 0) | \
            (engine.RUNOFF_KEEP_NAN if keep_nan else 0)
    got = engine.runoff_to_qlateral(W.indptr, W.indices, W.data, runoff, area if volumes else None, flags)
    want = oracle.runoff_to_qlateral_core(W, runoff, area if volumes else None, cumulative, clip, keep_nan)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    assert_close(np.nan_to_num(got), np.nan_to_num(want), 'qlateral')


def test_time_major_layout_through_the_raw_abi():
    """stride_p = 1 (the block as read from the file), the layout the python wrapper does not use."""
    from river_route_amd import _lib
    W, runoff, area = case(2000, 333, 50, np.float32, False, seed=4)
    out = np.empty((50, 2000))
    ip, ix, wd = W.indptr.astype(np.int32), W.indices.astype(np.int32), W.data.astype(np.float64)
    rc = _lib.lib().rr_runoff_to_qlateral(0, 2000, 333, 50, _lib.ptr(ip), _lib.ptr(ix), _lib.ptr(wd), _lib.ptr(runoff), 1,
                                          333, 1, _lib.ptr(area), 2, _lib.ptr(out))
    assert rc == 0
    assert_close(out, oracle.runoff_to_qlateral_core(W, runoff, area, False, True), 'time-major')


def test_large_block_and_bad_arguments():
    from river_route_amd import engine, _lib
    W, runoff, area = case(200_000, 30_000, 48, np.float32, False, seed=9)
    got = engine.runoff_to_qlateral(W.indptr, W.indices, W.data, runoff, area)
    assert_close(got, oracle.runoff_to_qlateral_core(W, runoff, area), '200k rivers')
    with pytest.raises(ValueError):
        engine.runoff_to_qlateral(W.indptr, W.indices, W.data, runoff[:, :100], area)
    assert _lib.lib().rr_runoff_to_qlateral(0, 10, 10, 5, None, None, None, None, 0, 1, 5, None, 0, None) < 0


@pytest.mark.parametrize('dtype,cumulative,clip', [(np.float32, False, False), (np.float64, True, True), (np.float32, True, False)])
@pytest.mark.parametrize('n_rivers,T,factor', [(60_000, 200, 1), (60_000, 256, 4)])
def test_routing_fed_by_gridded_runoff_in_one_call(n_rivers, T, factor, dtype, cumulative, clip):
    """rr_rapid_route_runoff_dev: runoff.py:288-332 + RapidMuskingum._router with the catchment inflow computed on its way into
    the engine's records, against the oracle's weights product followed by the oracle's rapid_route."""
    from river_route_amd import engine, synth
    from river_route_amd.engine import DeviceBuffer, Plan
    n_points = 9_000
    rng = np.random.default_rng(3)
    W = random_weights(rng, n_rivers, n_points)
    runoff = (rng.random((T, n_points)) * 2e-3 - 2e-4).astype(dtype)
    runoff[rng.integers(0, T, 9), rng.integers(0, n_points, 9)] = np.nan
    if cumulative:
        runoff = np.cumsum(np.nan_to_num(runoff), axis=0).astype(dtype)
    area = rng.uniform(1e5, 5e7, n_rivers)
    net = synth.synth_network(n_rivers, seed=6)
    has = net.down_index >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
    indices = net.down_index[has].astype(np.int32)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0)
    c4_dt = (c1 + c2) / 900.0
    ql = oracle.runoff_to_qlateral_core(W, runoff, area, cumulative, clip)
    q0 = 2.0 * synth.u01(3, np.arange(n_rivers))
    q_ref, d_ref = q0.copy(), np.zeros((T, n_rivers))
    oracle.rapid_route(indptr, indices, -c1[indices], c2, c3, c4_dt, q_ref, ql, d_ref, 1)
    flags = (engine.RUNOFF_CUMULATIVE if cumulative else 0) | (engine.RUNOFF_FORCE_POSITIVE if clip else 0)
    t_pad = -(-T // 16) * 16
    block = np.zeros((n_points, t_pad), dtype=dtype)
    block[:, :T] = runoff.T
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(-c1[indices], c2, c3, c4_dt)
        bufs = [DeviceBuffer(block.nbytes).upload(block), DeviceBuffer(W.indptr.nbytes).upload(W.indptr.astype(np.int32)),
                DeviceBuffer(W.indices.nbytes).upload(W.indices.astype(np.int32)), DeviceBuffer(W.data.nbytes).upload(W.data.astype(np.float64)),
                DeviceBuffer(area.nbytes).upload(area), DeviceBuffer(n_rivers * 8).upload(q0)]
        d_block, d_ptr, d_idx, d_w, d_area, d_q = bufs
        if factor == 1:
            d_out = DeviceBuffer(T * n_rivers * 8)
            plan.rapid_route_runoff_dev(d_q, n_points, d_ptr, d_idx, d_w, d_block, dtype == np.float32, 1, t_pad, d_area, flags, T, discharge=d_out)
            assert_close(d_out.download(np.float64, (T, n_rivers)), d_ref, 'discharge')
        else:
            d_out = DeviceBuffer((T // factor) * n_rivers * 4)
            plan.rapid_route_runoff_dev(d_q, n_points, d_ptr, d_idx, d_w, d_block, dtype == np.float32, 1, t_pad, d_area, flags, T,
                                        discharge32=d_out, factor=factor)
            want = d_ref.reshape(T // factor, factor, n_rivers).mean(axis=1)
            got = d_out.download(np.float32, (T // factor, n_rivers))
            np.testing.assert_allclose(got, want, rtol=3e-7, atol=1e-10 * np.abs(want).max())
        assert_close(d_q.download(np.float64, (n_rivers,)), q_ref, 'q_t')
        for b in bufs + [d_out]:
            b.free()

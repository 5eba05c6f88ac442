"""CPU-side checks of the product: the C-ABI library loads and exports every symbol of include/rr_hip.h, the
host-side network analysis (rr_plan.cpp) produces the layout the kernels assume, and nothing computes without a GPU."""
import os
import re

import numpy as np
import pytest

from conftest import REPO
from river_route_amd import _lib, synth
from river_route_amd.engine import Plan, RRError


def csc_from_down(down_index):
    has = down_index >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
    return indptr, down_index[has].astype(np.int32)


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, 'include', 'rr_hip.h')).read()
    declared = set(re.findall(r'\b(rr_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations parsed'
    lib = _lib.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), f'{name} declared in rr_hip.h but not exported'
    assert declared == set(_lib.EXPORTS), 'python binding table and header disagree'
    assert lib.rr_version() >= 100


@pytest.mark.parametrize('n,order', [(1, 'random'), (2, 'random'), (9, 'random'), (10, 'levels'), (1000, 'random'),
                                     (20001, 'random'), (5000, 'bfs')])
def test_plan_layout_invariants(n, order):
    net = synth.synth_network(n, order=order)
    indptr, indices = csc_from_down(net.down_index)
    with Plan(indptr, indices, device=_lib.RR_DEVICE_NONE) as plan:
        perm, lag, child_ptr = plan.layout()
        assert plan.n == n and plan.n_edges == len(indices)
        assert plan.depth == net.depth()
        assert sorted(perm.tolist()) == list(range(n))
        assert np.all(np.diff(lag) >= 0) and lag[0] == 0 and lag[-1] == plan.depth - 1
        assert child_ptr[0] == 0 and child_ptr[-1] == plan.n_edges and np.all(np.diff(child_ptr) >= 0)
        inv = np.empty(n, dtype=np.int64)
        inv[perm] = np.arange(n)
        # the reaches flowing into position p are exactly positions [child_ptr[p], child_ptr[p+1]), one level up
        if n <= 1000:
            for p in range(n):
                ups = np.flatnonzero(net.down_index == perm[p])
                assert sorted(inv[ups].tolist()) == list(range(child_ptr[p], child_ptr[p + 1]))
        down_pos = np.where(net.down_index[perm] >= 0, inv[np.maximum(net.down_index[perm], 0)], -1)
        has = down_pos >= 0
        assert np.all(lag[down_pos[has]] == lag[has] + 1)
        cnt = np.bincount(down_pos[has], minlength=n)
        assert np.array_equal(cnt, np.diff(child_ptr))
        assert plan.n_headwaters == int((cnt == 0).sum())
        if order == 'bfs':
            assert plan.identity_order


def test_plan_rejections_match_reference_messages():
    """tools.py:103-104 -> 'topologically sorted'; malformed CSC -> invalid."""
    with pytest.raises(RRError, match='topologically sorted') as e:
        Plan(np.array([0, 1, 1, 2], dtype=np.int32), np.array([1, 0], dtype=np.int32), device=_lib.RR_DEVICE_NONE)
    assert e.value.code == _lib.RR_E_NOT_TOPOLOGICAL
    with pytest.raises(RRError) as e:
        Plan(np.array([0, 1, 1], dtype=np.int32), np.array([7], dtype=np.int32), device=_lib.RR_DEVICE_NONE)
    assert e.value.code == _lib.RR_E_INVALID
    with pytest.raises(RRError) as e:   # two downstream reaches for reach 0
        Plan(np.array([0, 2, 2, 2], dtype=np.int32), np.array([1, 2], dtype=np.int32), device=_lib.RR_DEVICE_NONE)
    assert e.value.code == _lib.RR_E_UNSUPPORTED


def test_no_cpu_fallback():
    """A host-only plan must refuse to compute; so must a device plan when no GPU is visible."""
    net = synth.synth_network(50)
    indptr, indices = csc_from_down(net.down_index)
    with Plan(indptr, indices, device=_lib.RR_DEVICE_NONE) as plan:
        with pytest.raises(RRError) as e:
            plan.set_coeffs(np.zeros(plan.n_edges), np.zeros(50), np.zeros(50))
        assert e.value.code == _lib.RR_E_NO_DEVICE
        q = np.zeros(50)
        with pytest.raises(RRError) as e:
            plan.rapid_route(q, np.zeros((2, 50)), np.zeros((2, 50)), 1)
        assert e.value.code == _lib.RR_E_NO_DEVICE
    if _lib.device_count() == 0:
        with pytest.raises(RRError) as e:
            Plan(indptr, indices, device=0)
        assert e.value.code == _lib.RR_E_NO_DEVICE


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under river_route_amd/ may reference it."""
    pkg = os.path.join(REPO, 'river_route_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.cpp', '.hip', '.hpp', '.h')):
                text = open(os.path.join(root, f)).read()
                assert 'import oracle' not in text and 'from oracle' not in text and 'rr_oracle' not in text, f


def test_network_tools_digraph_and_subset(tmp_path):
    """river_route/tools.py:20-72: connectivity_to_digraph keeps the -1 sentinel edge (tests/test_tools.py:71-77 of the
    reference); subset_configs_to_river keeps the target and everything upstream and makes the target the outlet."""
    import pandas as pd
    from river_route_amd import synth, tools
    ids = np.array([10, 20, 30, 40], dtype=np.int64)
    down = np.array([30, 30, 40, -1], dtype=np.int64)
    g = tools.connectivity_to_digraph(ids, down)
    assert set(g.edges()) == {(10, 30), (20, 30), (30, 40), (40, -1)} and -1 in g.nodes
    net = synth.synth_network(2000, seed=12)
    table = pd.DataFrame({'river_id': net.river_ids, 'downstream_river_id': net.downstream_ids, 'k': net.k, 'x': net.x})
    params, out = tmp_path / 'params.parquet', tmp_path / 'subset.parquet'
    table.to_parquet(params)
    target = int(net.river_ids[1500])
    tools.subset_configs_to_river(target, params, out)
    sub = pd.read_parquet(out)
    import networkx as nx
    want = set(nx.ancestors(tools.connectivity_to_digraph(net.river_ids, net.downstream_ids), target)) | {target}
    assert set(sub['river_id']) == want
    assert int(sub.loc[sub['river_id'] == target, 'downstream_river_id'].iloc[0]) == -1
    assert (sub['downstream_river_id'][sub['river_id'] != target].isin(sub['river_id'])).all()
    tools.adjacency_matrix(sub['river_id'].to_numpy(), sub['downstream_river_id'].to_numpy())      # still sorted upstream -> downstream


def test_packed_runoff_with_fill_values_is_masked(tmp_path):
    """CF packing with a _FillValue (how ERA5-style runoff files store no-data cells): the reference reads through
    xarray's mask_and_scale, so fill cells are NaN (and end up as zero inflow), never fill * scale."""
    from scipy.io import netcdf_file
    from river_route_amd.io import read_variables
    path = tmp_path / 'packed.nc'
    raw = np.array([[100, -32767, 300], [-32767, 500, 600]], dtype=np.int16)
    with netcdf_file(str(path), 'w', version=2) as ds:
        ds.createDimension('time', 2)
        ds.createDimension('x', 3)
        v = ds.createVariable('ro', 'i2', ('time', 'x'))
        v[:] = raw
        v.scale_factor, v.add_offset, v._FillValue = 0.5, 10.0, np.int16(-32767)
        w = ds.createVariable('plain', 'f4', ('time', 'x'))
        w[:] = raw.astype(np.float32)
        w.missing_value = np.float32(-32767.0)
    got = read_variables(path, ['ro', 'plain'])
    ro, plain = got['ro'][0], got['plain'][0]
    want = np.where(raw == -32767, np.nan, raw * 0.5 + 10.0)
    np.testing.assert_array_equal(np.isnan(ro), raw == -32767)
    np.testing.assert_allclose(ro[~np.isnan(ro)], want[~np.isnan(want)])
    np.testing.assert_array_equal(np.isnan(plain), raw == -32767)
    assert plain.dtype == np.float32 and plain[0, 0] == 100.0


def test_torch_forcing_generator_matches_numpy():
    """bench.py makes its forcing on the device (synth.synth_qlateral_torch); the oracle legs use numpy's: the same bits."""
    from river_route_amd import synth
    n = 12_345
    a = synth.synth_qlateral(n, 3, 70, dt=450.0)
    b = synth.synth_qlateral_torch(n, 3, 70, 'cpu', dt=450.0).numpy()
    np.testing.assert_array_equal(a, b)
    cols = np.array([0, 17, 5000, n - 1])
    np.testing.assert_array_equal(synth.synth_qlateral_torch(n, 3, 70, 'cpu', columns=cols, dt=450.0).numpy(), a[:, cols])

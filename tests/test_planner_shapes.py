"""Adversarial network shapes through the host-side planners (rr_plan.cpp: lag layout, subtree tiles, direct row tiles, partitioner,
post-order) on host-only plans: what the index arithmetic must survive whatever the river network looks like.  tests/test_sanitize.py
runs this file (and the other host-only planner tests) again against the AddressSanitizer / UBSan build of the library."""
import numpy as np
import pytest

from river_route_amd import synth, tools
from river_route_amd._lib import RR_DEVICE_NONE, RRError
from river_route_amd.engine import Plan, partition_forest


def csc_from_down(down_index):
    has = down_index >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
    return indptr, down_index[has].astype(np.int32)


def check_tile_layout(down):
    """Every reach has exactly one position, every upstream edge is inside the tile or mirrored by a ghost, levels are consistent."""
    n = down.shape[0]
    indptr, indices = csc_from_down(down)
    with Plan(indptr, indices, device=RR_DEVICE_NONE) as plan:
        info = plan.tile_info()
        dinfo = plan.direct_info()
        if not info['ok']:
            return info, dinfo
        L = plan.tile_layout()
    GHOST = 1 << 28
    real = (L['lag'] & GHOST) == 0
    assert real.sum() == n and np.array_equal(np.sort(L['perm'][real]), np.arange(n))
    tile_of = np.repeat(np.arange(info['tiles']), np.diff(L['tile_ptr']))
    cnt = (L['ccnt'] & 0xFFFF).astype(np.int64)
    first = L['cfirst'].astype(np.int64)
    n_up = np.bincount(down[down >= 0], minlength=n)
    assert np.array_equal(cnt[real], n_up[L['perm'][real]])
    ends = first + cnt
    assert np.all(ends[real] <= L['tile_ptr'][tile_of[real] + 1]) and np.all(first[real][cnt[real] > 0] > np.flatnonzero(real)[cnt[real] > 0])
    return info, dinfo


def chain(n):
    d = np.arange(1, n + 1, dtype=np.int64)
    d[-1] = -1
    return d


def test_chain_ten_thousand_deep():
    info, dinfo = check_tile_layout(chain(10_000))
    assert info['ok'] and info['levels'] >= 10_000 // 512
    assert not dinfo['ok']      # every subtree taller than the window: records
    order = tools.postorder(np.arange(10_000)[::-1] + 5, np.concatenate([[-1], np.arange(10_000 - 1)[::-1] + 6]))      # rows downstream-first
    assert np.array_equal(order, np.arange(10_000)[::-1])


def test_thousand_way_confluence_streams():
    n = 1001
    down = np.full(n, n - 1, dtype=np.int64)
    down[-1] = -1
    info, dinfo = check_tile_layout(down)
    assert not info['ok'] and not dinfo['ok']      # more tributaries than a tile holds: the streaming kernel routes it
    part_of, sizes = partition_forest(*csc_from_down(down), 4)
    assert sizes.sum() == n


def test_forest_of_singletons_and_pairs():
    n = 100_000
    down = np.full(n, -1, dtype=np.int64)
    down[0:n:2] = np.arange(1, n, 2)      # half of them pairs
    info, dinfo = check_tile_layout(down)
    assert info['ok'] and info['levels'] == 1 and info['ghosts'] == 0
    assert dinfo['ok'] and dinfo['holes'] == 0 and dinfo['tiles'] == -(-n // 256)
    part_of, sizes = partition_forest(*csc_from_down(down), 8)
    assert sizes.sum() == n and np.all(part_of[0:n:2] == part_of[1:n:2]), 'a pair is never cut'


def test_comb_with_a_long_stem():
    m = 40_000
    down = np.concatenate([m + np.arange(m), m + 1 + np.arange(m)]).astype(np.int64)      # m teeth into a stem of m reaches
    down[-1] = -1
    info, dinfo = check_tile_layout(down)
    assert info['ok']
    part_of, sizes = partition_forest(*csc_from_down(down), 8)
    assert sizes.sum() == 2 * m and sizes.max() <= 1.3 * 2 * m / 8


@pytest.mark.parametrize('n', [1, 2, 3, 255, 256, 257, 511, 512, 513, 65_535, 65_537])
def test_sizes_around_the_tile_boundaries(n):
    for net in (synth.synth_network(n, seed=n, order='postorder'), synth.synth_network(n, seed=n + 1, order='random')):
        info, dinfo = check_tile_layout(net.down_index)
        assert info['ok']
        order = tools.postorder(net.river_ids, net.downstream_ids)
        assert np.array_equal(np.sort(order), np.arange(n))


def test_deep_binary_caterpillar_and_three_way_confluences():
    net = synth.synth_network_chain(150_000, p_chain=0.995, n_outlets=3, p_third=0.3, seed=3, order='postorder')
    info, dinfo = check_tile_layout(net.down_index)
    assert info['ok']
    net = synth.synth_network_chain(50_000, p_chain=0.0, n_outlets=500, p_third=0.5, seed=4, order='random')
    info, dinfo = check_tile_layout(net.down_index)
    assert info['ok']


def test_malformed_structures_are_refused_not_walked():
    with pytest.raises(RRError):      # row <= column: not topologically sorted
        Plan(np.array([0, 1, 2], dtype=np.int32), np.array([1, 0], dtype=np.int32), device=RR_DEVICE_NONE)
    with pytest.raises(RRError):      # row index out of range
        Plan(np.array([0, 1, 1], dtype=np.int32), np.array([7], dtype=np.int32), device=RR_DEVICE_NONE)
    with pytest.raises(RRError):      # two downstream reaches
        Plan(np.array([0, 2, 2, 2], dtype=np.int32), np.array([1, 2], dtype=np.int32), device=RR_DEVICE_NONE)
    with pytest.raises(RRError):      # indptr decreasing
        Plan(np.array([0, 1, 0, 1], dtype=np.int32), np.array([2], dtype=np.int32), device=RR_DEVICE_NONE)

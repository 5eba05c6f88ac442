"""Host-side invariants of the subtree-tile layout the time-tiled kernel runs on (rr::TilePlan).
No GPU needed: plans are host-only."""
import numpy as np
import pytest

from river_route_amd import synth
from river_route_amd._lib import RR_DEVICE_NONE, RRError
from river_route_amd.engine import Plan

GHOST, EXPORT, MASK = 1 << 28, 1 << 27, (1 << 27) - 1


def csc_from_down(down_index):
    has = down_index >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
    return indptr, down_index[has].astype(np.int32)


def check_layout(down, block):
    n = down.shape[0]
    indptr, indices = csc_from_down(down)
    with Plan(indptr, indices, device=RR_DEVICE_NONE) as plan:
        info = plan.tile_info()
        assert info['ok'] and info['block'] == block
        L = plan.tile_layout()
        _, lag_h, _ = plan.layout()
        perm_h = plan.layout()[0]
    lag_of = np.empty(n, np.int64)
    lag_of[perm_h] = lag_h
    tile_ptr, level, perm, lag, cfirst, ccnt, xpos = (L[k] for k in ('tile_ptr', 'tile_level', 'perm', 'lag', 'cfirst', 'ccnt', 'xpos'))
    npos = info['positions']
    assert tile_ptr[0] == 0 and tile_ptr[-1] == npos and np.all(np.diff(tile_ptr) > 0) and np.diff(tile_ptr).max() <= block
    assert np.all(np.diff(level) >= 0) and level[0] == 0 and level[-1] == info['levels'] - 1
    ghost = (lag & GHOST) != 0
    assert ghost.sum() == info['ghosts'] and npos == n + info['ghosts']
    # every reach has exactly one real position; a position's lag is its reach's lag
    real_pos = np.flatnonzero(~ghost)
    assert np.array_equal(np.sort(perm[real_pos]), np.arange(n))
    assert np.array_equal(lag & MASK, lag_of[perm])
    inv = np.empty(n, np.int64)
    inv[perm[real_pos]] = real_pos
    tile_of = np.repeat(np.arange(tile_ptr.size - 1), np.diff(tile_ptr))
    # upstream positions: contiguous, in the same tile, exactly the reaches that flow into the position's reach
    n_up = np.bincount(down[down >= 0], minlength=n)
    cnt, hw = (ccnt & 0xFFFF).astype(np.int64), (ccnt >> 16).astype(np.int64)
    assert np.all(cnt[ghost] == 0) and np.array_equal(cnt[real_pos], n_up[perm[real_pos]])
    edges_p = np.repeat(np.arange(npos), cnt)
    edges_u = np.concatenate([cfirst[p] + np.arange(c) for p, c in zip(real_pos, cnt[real_pos]) if c]) if cnt.sum() else np.zeros(0, np.int64)
    assert np.array_equal(down[perm[edges_u]], perm[edges_p])
    assert np.array_equal(tile_of[edges_u], tile_of[edges_p])
    # headwater tributaries come first
    rank = np.concatenate([np.arange(c) for c in cnt[real_pos] if c]) if cnt.sum() else np.zeros(0, np.int64)
    assert np.array_equal(n_up[perm[edges_u]] == 0, rank < hw[edges_p])
    # a ghost mirrors a reach of a LOWER tile level, which carries the export flag and points back at it
    g = np.flatnonzero(ghost)
    src = inv[perm[g]]
    assert np.all(level[tile_of[src]] < level[tile_of[g]])
    assert np.all((lag[src] & EXPORT) != 0) and np.array_equal(xpos[src], g) and np.array_equal(xpos[g], src)
    assert (lag & EXPORT).astype(bool).sum() == g.size
    return info


@pytest.mark.parametrize('n,block,seed', [(300, 512, 1), (5000, 64, 2), (60000, 512, 21), (60000, 40, 5), (250000, 512, 13)])
def test_random_networks(monkeypatch, n, block, seed):
    monkeypatch.setenv('RR_TILE_BLOCK', str(block))
    info = check_layout(synth.synth_network(n, seed=seed).down_index, block)
    if n >= 60000 and block == 512:
        assert info['tiles'] <= 1.06 * n / block + 8       # tiles are full: the skeleton costs a few per cent
        assert info["levels"] <= 24                        # the schedule's skew is levels x K ticks (a chain of blocks would be n / block)


def test_degenerate_shapes(monkeypatch):
    monkeypatch.setenv('RR_TILE_BLOCK', '128')
    n = 3000
    chain = np.arange(1, n + 1, dtype=np.int64)
    chain[-1] = -1
    info = check_layout(chain, 128)
    assert info['levels'] >= n // 128            # a chain is serial: one level per tile
    lone = np.full(777, -1, dtype=np.int64)
    assert check_layout(lone, 128)['levels'] == 1
    m = 1500   # comb: reaches 0..m-1 are tributaries, m..2m-1 the stem
    comb = np.concatenate([m + np.arange(m), m + 1 + np.arange(m)]).astype(np.int64)
    comb[-1] = -1
    check_layout(comb, 128)
    fan = np.concatenate([np.full(100, 100), 100 + 1 + np.arange(60)]).astype(np.int64)   # 100 tributaries into one reach
    fan[-1] = -1
    check_layout(fan, 128)


def test_too_many_upstream_reaches_for_a_tile(monkeypatch):
    monkeypatch.setenv('RR_TILE_BLOCK', '128')
    star = np.full(501, 500, dtype=np.int64)
    star[-1] = -1
    indptr, indices = csc_from_down(star)
    with Plan(indptr, indices, device=RR_DEVICE_NONE) as plan:
        assert not plan.tile_info()['ok']          # routed by the streaming kernel
        with pytest.raises(RRError):
            plan.tile_layout()

"""
Pins the CPU oracle (oracle/rr_oracle.c) against golden vectors produced by running the reference
(tests/golden/make_golden.py) and against the known-answer cases the reference's own tests hold.
"""
import numpy as np
import pytest

from conftest import assert_close, unit_split
from oracle import oracle

CASES = [('docs9', (1, 4), (1, 3)), ('tree1k', (1, 4), (3,)), ('forest30', (1, 3), (48,))]


@pytest.mark.parametrize('tag,nsubs,n_ks_list', CASES)
def test_adjacency_and_coefficients(golden_kernels, tag, nsubs, n_ks_list):
    g = golden_kernels
    indptr, indices = oracle.adjacency_csc(g[f'{tag}/river_ids'], g[f'{tag}/downstream_ids'])
    np.testing.assert_array_equal(indptr, g[f'{tag}/indptr'])
    np.testing.assert_array_equal(indices, g[f'{tag}/indices'])
    c1, c2, c3 = oracle.muskingum_coefficients(g[f'{tag}/k'], g[f'{tag}/x'], float(g[f'{tag}/dt']))
    # same IEEE operations in the same order as Muskingum.py:174-179 -> bit-exact
    np.testing.assert_array_equal(c1, g[f'{tag}/c1'])
    np.testing.assert_array_equal(c2, g[f'{tag}/c2'])
    np.testing.assert_array_equal(c3, g[f'{tag}/c3'])
    np.testing.assert_array_equal(-c1[indices], g[f'{tag}/lhs_off'])


@pytest.mark.parametrize('tag,nsubs,n_ks_list', CASES)
def test_rapid_and_muskingum_route(golden_kernels, tag, nsubs, n_ks_list):
    g = golden_kernels
    args = [g[f'{tag}/{k}'] for k in ('indptr', 'indices', 'lhs_off', 'c2', 'c3')]
    ql = g[f'{tag}/qlateral']
    T, n = ql.shape
    for nsub in nsubs:
        q_t = g[f'{tag}/q0'].copy()
        d = np.zeros((T, n))
        oracle.rapid_route(*args, g[f'{tag}/rapid{nsub}/c4_dt'], q_t, ql, d, nsub)
        assert_close(q_t, g[f'{tag}/rapid{nsub}/q_t'], f'{tag} rapid{nsub} q_t')
        assert_close(d, g[f'{tag}/rapid{nsub}/discharge'], f'{tag} rapid{nsub} discharge')
        n_out = max(T // 4, 1)
        q_t = g[f'{tag}/q0'].copy()
        d = np.zeros((n_out, n))
        oracle.muskingum_route(*args, q_t, d, n_out, nsub)
        assert_close(q_t, g[f'{tag}/musk{nsub}/q_t'], f'{tag} musk{nsub} q_t')
        assert_close(d, g[f'{tag}/musk{nsub}/discharge'], f'{tag} musk{nsub} discharge')


@pytest.mark.parametrize('tag,nsubs,n_ks_list', CASES)
def test_unit_route_and_convolution(golden_kernels, tag, nsubs, n_ks_list):
    g = golden_kernels
    indptr, indices = g[f'{tag}/indptr'], g[f'{tag}/indices']
    n = len(indptr) - 1
    hw_idx, inner_idx, A_in, A_hw = unit_split(indptr, indices, n)
    np.testing.assert_array_equal(hw_idx, g[f'{tag}/hw_idx'])
    np.testing.assert_array_equal(inner_idx, g[f'{tag}/inner_idx'])
    c1i, c2i, c3i = (g[f'{tag}/{c}'][inner_idx] for c in ('c1', 'c2', 'c3'))
    lhs_in = np.ascontiguousarray(-c1i[A_in.indices])
    for n_ks in n_ks_list:
        p = f'{tag}/unit_ks{n_ks}'
        uh = oracle.UnitHydrograph(g[f'{p}/kernel'])
        uh.state = g[f'{p}/state0'].copy()
        conv = uh.convolve(g[f'{p}/depth'])
        # direct form vs the reference's FFT evaluation: rounding-level agreement only
        scale = np.abs(g[f'{p}/convolved']).max()
        np.testing.assert_allclose(conv, g[f'{p}/convolved'], rtol=0, atol=1e-12 * scale)
        np.testing.assert_allclose(uh.state, g[f'{p}/state1'], rtol=0, atol=1e-12 * scale)
        for nsub in nsubs:
            q_ch = g[f'{tag}/q0'][inner_idx].copy()
            q_full = q_ch.copy()
            d = np.zeros_like(g[f'{p}/depth'])
            oracle.unit_route(A_in.indptr, A_in.indices, lhs_in, A_in.indptr, A_in.indices, A_in.data,
                              A_hw.indptr, A_hw.indices, A_hw.data, c1i, c2i, c3i, hw_idx, inner_idx,
                              q_ch, q_full, g[f'{p}/convolved'], d, nsub)
            assert_close(q_ch, g[f'{p}/nsub{nsub}/q_ch'], f'{p} nsub{nsub} q_ch')
            assert_close(q_full, g[f'{p}/nsub{nsub}/q_full'], f'{p} nsub{nsub} q_full')
            assert_close(d, g[f'{p}/nsub{nsub}/discharge'], f'{p} nsub{nsub} discharge')


@pytest.mark.parametrize('ci', range(5))
def test_convolve_golden(golden_kernels, ci):
    g = golden_kernels
    uh = oracle.UnitHydrograph(g[f'conv{ci}/kernel'])
    uh.state = g[f'conv{ci}/state0'].copy()
    for leg in 'ab':
        got = uh.convolve(g[f'conv{ci}/lat_{leg}'])
        np.testing.assert_allclose(got, g[f'conv{ci}/out_{leg}'], rtol=0, atol=1e-12)
        np.testing.assert_allclose(uh.state, g[f'conv{ci}/state_{leg}'], rtol=0, atol=1e-12)
    uh2 = oracle.UnitHydrograph(g[f'conv{ci}/kernel'])
    lat = np.vstack([g[f'conv{ci}/lat_a'], g[f'conv{ci}/lat_b']])
    inc = np.stack([uh2.convolve_incrementally(r) for r in lat])
    np.testing.assert_allclose(inc, g[f'conv{ci}/incremental_zero_state'], rtol=1e-14, atol=0)


def test_convolve_vs_convolve_incrementally():
    """Known-answer restatement of the reference's tests/test_uhkernels.py:52-78 (seed 123, rtol 1e-12)."""
    np.random.seed(123)
    kernel = np.random.rand(3, 4)
    lateral = np.random.rand(10, 4)
    full = oracle.UnitHydrograph(kernel).convolve(lateral)
    uh = oracle.UnitHydrograph(kernel)
    inc = np.stack([uh.convolve_incrementally(lateral[t]) for t in range(10)])
    np.testing.assert_allclose(full, inc, rtol=1e-12)


def test_convolve_impulse_response():
    """tests/test_uhkernels.py:81-99: a unit impulse reproduces the kernel columns, then zeros."""
    kernel = np.array([[1.0, 0.5], [0.5, 0.3], [0.0, 0.2]])
    lateral = np.zeros((5, 2))
    lateral[0, :] = 1.0
    res = oracle.UnitHydrograph(kernel).convolve(lateral)
    np.testing.assert_allclose(res[:3], kernel, rtol=1e-12)
    np.testing.assert_allclose(res[3:], 0.0, atol=1e-15)


def test_adjacency_rejections():
    """tests/test_tools.py:48-60."""
    with pytest.raises(ValueError, match='topologically sorted'):
        oracle.adjacency_csc(np.array([10, 20, 30]), np.array([20, -1, 10]))
    with pytest.raises(ValueError, match='Unknown downstream_river_id'):
        oracle.adjacency_csc(np.array([10, 20]), np.array([-1, 999]))


def test_coefficients_must_sum_to_one(golden_kernels):
    """Muskingum.py:180-185: k = 0 makes the coefficients NaN -> ValueError with the reference's message."""
    with pytest.raises(ValueError) as e:
        oracle.muskingum_coefficients(np.array([3600.0, 0.0]), np.array([0.2, 0.2]), 900.0)
    assert str(e.value) == str(golden_kernels['coeff_fail/message'])


def test_zero_state_gives_zero_output(golden_kernels):
    """tests/test_muskingum.py:48-71."""
    g = golden_kernels
    n = len(g['tree1k/indptr']) - 1
    q_t = np.zeros(n)
    d = np.ones((3, n))
    oracle.muskingum_route(g['tree1k/indptr'], g['tree1k/indices'], g['tree1k/lhs_off'], g['tree1k/c2'],
                           g['tree1k/c3'], q_t, d, 3, 2)
    assert not d.any() and not q_t.any()

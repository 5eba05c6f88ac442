"""
GPU parity tests: the HIP engine, called through the C ABI exactly as the reference calls its numba kernels
(river_route_amd.kernels mirrors river_route/routers/_numba_kernels.py), against
  (a) the golden vectors produced by running the reference (tests/golden/kernels.npz),
  (b) the CPU oracle on seeded inputs at sizes it finishes in seconds,
  (c) size-independent properties at larger sizes.
Tolerance: fp64 rtol 1e-10, atol 1e-10 * max|Q|.
"""
import numpy as np
import pytest

from conftest import assert_close, unit_split
from oracle import oracle
from river_route_amd import kernels, synth
from river_route_amd.engine import Plan, uh_convolve

pytestmark = pytest.mark.gpu

CASES = [('docs9', (1, 4), (1, 3)), ('tree1k', (1, 4), (3,)), ('forest30', (1, 3), (48,))]


def csc_from_down(down_index):
    has = down_index >= 0
    indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
    return indptr, down_index[has].astype(np.int32)


def coeffs(net, dt):
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, dt)
    return c1, c2, c3


# ---------------------------------------------------------------- (a) golden vectors from the reference

@pytest.mark.parametrize('tag,nsubs,n_ks_list', CASES)
def test_rapid_and_muskingum_golden(golden_kernels, tag, nsubs, n_ks_list):
    g = golden_kernels
    args = [g[f'{tag}/{k}'] for k in ('indptr', 'indices', 'lhs_off', 'c2', 'c3')]
    ql = g[f'{tag}/qlateral']
    T, n = ql.shape
    for nsub in nsubs:
        q_t = g[f'{tag}/q0'].copy()
        d = np.zeros((T, n))
        kernels.rapid_route(*args, g[f'{tag}/rapid{nsub}/c4_dt'], q_t, ql, d, nsub)
        assert_close(q_t, g[f'{tag}/rapid{nsub}/q_t'], f'{tag} rapid{nsub} q_t')
        assert_close(d, g[f'{tag}/rapid{nsub}/discharge'], f'{tag} rapid{nsub} discharge')
        n_out = max(T // 4, 1)
        q_t = g[f'{tag}/q0'].copy()
        d = np.zeros((n_out, n))
        kernels.muskingum_route(*args, q_t, d, n_out, nsub)
        assert_close(q_t, g[f'{tag}/musk{nsub}/q_t'], f'{tag} musk{nsub} q_t')
        assert_close(d, g[f'{tag}/musk{nsub}/discharge'], f'{tag} musk{nsub} discharge')


@pytest.mark.parametrize('tag,nsubs,n_ks_list', CASES)
def test_unit_route_and_convolution_golden(golden_kernels, tag, nsubs, n_ks_list):
    g = golden_kernels
    indptr, indices = g[f'{tag}/indptr'], g[f'{tag}/indices']
    n = len(indptr) - 1
    hw_idx, inner_idx, A_in, A_hw = unit_split(indptr, indices, n)
    c1i, c2i, c3i = (g[f'{tag}/{c}'][inner_idx] for c in ('c1', 'c2', 'c3'))
    lhs_in = np.ascontiguousarray(-c1i[A_in.indices])
    for n_ks in n_ks_list:
        p = f'{tag}/unit_ks{n_ks}'
        state = g[f'{p}/state0'].copy()
        conv = uh_convolve(g[f'{p}/kernel'], state, g[f'{p}/depth'])
        scale = np.abs(g[f'{p}/convolved']).max()
        # the reference evaluates the same linear convolution by FFT: agreement to rounding of the transform
        np.testing.assert_allclose(conv, g[f'{p}/convolved'], rtol=0, atol=1e-12 * scale)
        np.testing.assert_allclose(state, g[f'{p}/state1'], rtol=0, atol=1e-12 * scale)
        for nsub in nsubs:
            q_ch = g[f'{tag}/q0'][inner_idx].copy()
            q_full = q_ch.copy()
            d = np.zeros_like(g[f'{p}/depth'])
            kernels.unit_route(A_in.indptr, A_in.indices, lhs_in, A_in.indptr, A_in.indices, A_in.data,
                               A_hw.indptr, A_hw.indices, A_hw.data, c1i, c2i, c3i, hw_idx, inner_idx,
                               q_ch, q_full, g[f'{p}/convolved'], d, nsub)
            assert_close(q_ch, g[f'{p}/nsub{nsub}/q_ch'], f'{p} nsub{nsub} q_ch')
            assert_close(q_full, g[f'{p}/nsub{nsub}/q_full'], f'{p} nsub{nsub} q_full')
            assert_close(d, g[f'{p}/nsub{nsub}/discharge'], f'{p} nsub{nsub} discharge')


@pytest.mark.parametrize('ci', range(5))
def test_convolve_golden(golden_kernels, ci):
    """UnitHydrograph.convolve incl. T < n_ks, n_ks = 1 and state carried across two calls."""
    g = golden_kernels
    state = g[f'conv{ci}/state0'].copy()
    for leg in 'ab':
        got = uh_convolve(g[f'conv{ci}/kernel'], state, g[f'conv{ci}/lat_{leg}'])
        np.testing.assert_allclose(got, g[f'conv{ci}/out_{leg}'], rtol=0, atol=1e-12)
        np.testing.assert_allclose(state, g[f'conv{ci}/state_{leg}'], rtol=0, atol=1e-12)


def test_convolve_impulse_response():
    """tests/test_uhkernels.py:81-99 of the reference."""
    kernel = np.array([[1.0, 0.5], [0.5, 0.3], [0.0, 0.2]])
    lateral = np.zeros((5, 2))
    lateral[0, :] = 1.0
    res = uh_convolve(kernel, np.zeros_like(kernel), lateral)
    np.testing.assert_allclose(res[:3], kernel, rtol=1e-12)
    np.testing.assert_allclose(res[3:], 0.0, atol=1e-15)


# ---------------------------------------------------------------- (b) oracle on seeded inputs

@pytest.mark.parametrize('n,T,nsub,order', [(1, 5, 1, 'random'), (2, 7, 2, 'random'), (777, 70, 1, 'random'),
                                            (20000, 96, 1, 'random'), (20000, 24, 3, 'levels'),
                                            (20000, 24, 1, 'bfs'), (100000, 48, 1, 'random')])
def test_rapid_vs_oracle(n, T, nsub, order):
    net = synth.synth_network(n, order=order)
    indptr, indices = csc_from_down(net.down_index)
    dt = 900.0
    c1, c2, c3 = coeffs(net, dt)
    lhs = -c1[indices]
    c4_dt = (c1 + c2) / (dt * nsub)
    ql = synth.synth_qlateral(n, 0, T, dt=dt * nsub)
    q0 = 5.0 * synth.u01(99, np.arange(n))
    q_ref, d_ref = q0.copy(), np.zeros((T, n))
    oracle.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_ref, ql, d_ref, nsub)
    q_t, d = q0.copy(), np.zeros((T, n))
    kernels.rapid_route(indptr, indices, lhs, c2, c3, c4_dt, q_t, ql, d, nsub)
    assert_close(q_t, q_ref, 'q_t')
    assert_close(d, d_ref, 'discharge')


@pytest.mark.parametrize('n,n_out,nrpo', [(3000, 40, 1), (3000, 10, 4)])
def test_muskingum_vs_oracle(n, n_out, nrpo):
    net = synth.synth_network(n, seed=5)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = coeffs(net, 900.0)
    lhs = -c1[indices]
    q0 = 10.0 * synth.u01(3, np.arange(n))
    q_ref, d_ref = q0.copy(), np.zeros((n_out, n))
    oracle.muskingum_route(indptr, indices, lhs, c2, c3, q_ref, d_ref, n_out, nrpo)
    q_t, d = q0.copy(), np.zeros((n_out, n))
    kernels.muskingum_route(indptr, indices, lhs, c2, c3, q_t, d, n_out, nrpo)
    assert_close(q_t, q_ref, 'q_t')
    assert_close(d, d_ref, 'discharge')
    assert d.min() >= 0.0   # tests/test_muskingum.py:41


@pytest.mark.parametrize('n,T,nsub,n_ks', [(5000, 30, 1, 48), (5000, 12, 3, 7)])
def test_unit_vs_oracle(n, T, nsub, n_ks):
    net = synth.synth_network(n, seed=11)
    indptr, indices = csc_from_down(net.down_index)
    hw_idx, inner_idx, A_in, A_hw = unit_split(indptr, indices, n)
    c1, c2, c3 = coeffs(net, 900.0)
    c1i, c2i, c3i = c1[inner_idx], c2[inner_idx], c3[inner_idx]
    lhs_in = np.ascontiguousarray(-c1i[A_in.indices])
    kern = synth.synth_uh_kernel(n, n_ks)
    depth = synth.synth_runoff_depth(n, 0, T)
    st_ref = 0.1 * kern.copy()
    uh = oracle.UnitHydrograph(kern)
    uh.state = st_ref
    conv_ref = uh.convolve(depth)
    st = 0.1 * kern.copy()
    conv = uh_convolve(kern, st, depth)
    assert_close(conv, conv_ref, 'convolved')
    assert_close(st, uh.state, 'uh state')
    q0 = 5.0 * synth.u01(17, np.arange(n))
    args = (A_in.indptr, A_in.indices, lhs_in, A_in.indptr, A_in.indices, A_in.data,
            A_hw.indptr, A_hw.indices, A_hw.data, c1i, c2i, c3i, hw_idx, inner_idx)
    qc_ref, qf_ref, d_ref = q0[inner_idx].copy(), q0[inner_idx].copy(), np.zeros((T, n))
    oracle.unit_route(*args, qc_ref, qf_ref, conv_ref, d_ref, nsub)
    qc, qf, d = q0[inner_idx].copy(), q0[inner_idx].copy(), np.zeros((T, n))
    kernels.unit_route(*args, qc, qf, conv_ref, d, nsub)
    assert_close(qc, qc_ref, 'q_ch')
    assert_close(qf, qf_ref, 'q_full')
    assert_close(d, d_ref, 'discharge')


@pytest.mark.parametrize('n,T,nsub', [(4000, 20, 1), (4000, 9, 3), (9, 6, 2)])
def test_unit_route_general_edge_data_vs_oracle(n, T, nsub):
    """unit_route multiplies by a_inner_data[j] / a_hw_data[j] and subtracts lhs_off_data[j] q_ch
    (_numba_kernels.py:126-139, 159-162).  The reference's router passes ones and -c1[row]; other values are legal at the
    kernel boundary and go through the streaming kernel's general branch; afterwards the same plan routes unit weights
    with the time-tiled kernel again."""
    net = synth.synth_network(n, seed=13)
    indptr, indices = csc_from_down(net.down_index)
    hw_idx, inner_idx, A_in, A_hw = unit_split(indptr, indices, n)
    c1, c2, c3 = coeffs(net, 900.0 / nsub)
    c1i, c2i, c3i = c1[inner_idx], c2[inner_idx], c3[inner_idx]
    rng = np.random.default_rng(5)
    a_in, a_hw = rng.uniform(0.5, 1.5, A_in.data.size), rng.uniform(0.5, 1.5, A_hw.data.size)
    lhs_in = -c1i[A_in.indices] * rng.uniform(0.7, 1.2, A_in.data.size)
    conv = 3.0 * synth.synth_runoff_depth(n, 0, T) * 1e3
    q0 = 5.0 * synth.u01(17, np.arange(n))
    for weights in ((lhs_in, a_in, a_hw), (np.ascontiguousarray(-c1i[A_in.indices]), A_in.data, A_hw.data)):
        args = (A_in.indptr, A_in.indices, weights[0], A_in.indptr, A_in.indices, weights[1],
                A_hw.indptr, A_hw.indices, weights[2], c1i, c2i, c3i, hw_idx, inner_idx)
        qc_ref, qf_ref, d_ref = q0[inner_idx].copy(), 1.1 * q0[inner_idx], np.zeros((T, n))
        oracle.unit_route(*args, qc_ref, qf_ref, conv, d_ref, nsub)
        qc, qf, d = q0[inner_idx].copy(), 1.1 * q0[inner_idx], np.zeros((T, n))
        kernels.unit_route(*args, qc, qf, conv, d, nsub)
        assert_close(qc, qc_ref, 'q_ch')
        assert_close(qf, qf_ref, 'q_full')
        assert_close(d, d_ref, 'discharge')


# ---------------------------------------------------------------- (c) properties the reference's tests assert

def _rapid_setup(n, seed=NotImplemented):
    net = synth.synth_network(n) if seed is NotImplemented else synth.synth_network(n, seed=seed)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = coeffs(net, 900.0)
    plan = Plan(indptr, indices)
    plan.set_coeffs(-c1[indices], c2, c3, (c1 + c2) / 900.0)
    return net, plan


def test_split_run_equals_joint_run():
    """State round trip: month 1 -> state -> month 2 equals months 1+2 (tests/test_rapid_muskingum.py:95-143)."""
    n, T = 50000, 60
    net, plan = _rapid_setup(n)
    ql = synth.synth_qlateral(n, 0, T)
    q_joint, d_joint = np.zeros(n), np.zeros((T, n))
    plan.rapid_route(q_joint, ql, d_joint, 1)
    q_split, d1, d2 = np.zeros(n), np.zeros((25, n)), np.zeros((T - 25, n))
    plan.rapid_route(q_split, ql[:25], d1, 1)
    plan.rapid_route(q_split, ql[25:], d2, 1)
    # identical arithmetic in both runs -> bit-exact
    np.testing.assert_array_equal(np.vstack([d1, d2]), d_joint)
    np.testing.assert_array_equal(q_split, q_joint)
    plan.close()


def test_zero_in_zero_out_and_linearity_1m():
    """Full BASELINE size (1M reaches): zero state + zero lateral => exactly zero (tests/test_muskingum.py:48-71);
    routing is linear in (state, lateral) while nothing is clamped."""
    n, T = 1_000_000, 16
    net, plan = _rapid_setup(n)
    q, d = np.zeros(n), np.ones((T, n))
    plan.rapid_route(q, np.zeros((T, n)), d, 1)
    assert not q.any() and not d.any()
    ql_a = synth.synth_qlateral(n, 0, T)
    ql_b = synth.synth_qlateral(n, 100, 100 + T)
    outs = []
    for ql in (ql_a, ql_b, 2.0 * ql_a + 0.5 * ql_b):
        q, d = np.zeros(n), np.zeros((T, n))
        plan.rapid_route(q, ql, d, 1)
        outs.append((q, d))
    assert outs[0][1].min() >= 0.0
    assert_close(outs[2][0], 2.0 * outs[0][0] + 0.5 * outs[1][0], 'linearity q_t')
    # outputs are clamped at 0; with non-negative forcing and c's in range the clamp is rarely active, compare
    # where both summands are positive
    lin = 2.0 * outs[0][1] + 0.5 * outs[1][1]
    m = (outs[0][1] > 0) & (outs[1][1] > 0)
    assert m.mean() > 0.9
    assert_close(np.where(m, outs[2][1], 0.0), np.where(m, lin, 0.0), 'linearity discharge')
    # mass balance at steady forcing is covered by the oracle comparison at 100k; here check the oracle on a
    # sampled window of the 1M run instead: first 2 steps
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = coeffs(net, 900.0)
    q_ref, d_ref = np.zeros(n), np.zeros((2, n))
    oracle.rapid_route(indptr, indices, -c1[indices], c2, c3, (c1 + c2) / 900.0, q_ref, ql_a[:2], d_ref, 1)
    assert_close(outs[0][1][:2], d_ref, '1M first rows vs oracle')
    plan.close()


def test_initial_state_decays():
    """tests/test_rapid_muskingum.py:46-92: an initial state changes the first step and its influence decays."""
    n, T = 20000, 400
    net, plan = _rapid_setup(n, seed=3)
    ql = synth.synth_qlateral(n, 0, T)
    q_a, d_a = np.zeros(n), np.zeros((T, n))
    plan.rapid_route(q_a, ql, d_a, 1)
    q_b, d_b = np.full(n, 50.0), np.zeros((T, n))
    plan.rapid_route(q_b, ql, d_b, 1)
    # headwater reaches see only their own state: q+ = c3 q + c4dt ql, so the difference decays like c3^t;
    # downstream reaches first accumulate the extra water, so the check is made where the reference's claim is sharp
    hw = np.setdiff1d(np.arange(n), net.down_index[net.down_index >= 0])
    diff = np.abs(d_b - d_a)[:, hw].max(axis=1)
    assert diff[0] > 1.0
    assert diff[-1] < 1e-6
    assert np.abs(d_b - d_a).max(axis=1)[0] > 1.0
    plan.close()


@pytest.mark.parametrize('n,T,n_ks', [(3000, 300, 48), (700, 1000, 5), (129, 64, 64), (5000, 200, 17), (40, 70, 33)])
def test_long_series_convolution_vs_oracle(n, T, n_ks):
    """The register/LDS-window kernel (T >= 64, n_ks <= 64), several time segments, carried state in and out."""
    kern = synth.synth_uh_kernel(n, n_ks)
    depth = synth.synth_runoff_depth(n, 0, T)
    uh = oracle.UnitHydrograph(kern)
    uh.state = 0.3 * kern[::-1].copy()
    st = uh.state.copy()
    ref_a = uh.convolve(depth[:T // 2])
    ref_b = uh.convolve(depth[T // 2:])
    got_a = uh_convolve(kern, st, depth[:T // 2])
    got_b = uh_convolve(kern, st, depth[T // 2:])
    assert_close(got_a, ref_a, 'first half')
    assert_close(got_b, ref_b, 'second half (carried state)')
    assert_close(st, uh.state, 'state')


def test_unit_muskingum_1m_properties():
    """BASELINE config 4 size: 1M reaches, 48-step kernel.  Headwater discharge equals the convolved lateral exactly
    (_numba_kernels.py:122-123), outputs non-negative on inner reaches, first rows match the oracle."""
    from conftest import unit_split
    n, T, n_ks = 1_000_000, 12, 48
    net = synth.synth_network(n)
    indptr, indices = csc_from_down(net.down_index)
    c1, c2, c3 = coeffs(net, 900.0)
    kern = synth.synth_uh_kernel(n, n_ks)
    depth = synth.synth_runoff_depth(n, 0, T)
    st = np.zeros_like(kern)
    conv = uh_convolve(kern, st, depth)
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(-c1[indices], c2, c3, None)
        q_ch, q_full, d = np.zeros(plan.n_inner), np.zeros(plan.n_inner), np.zeros((T, n))
        plan.unit_route(q_ch, q_full, conv, d, 1)
    hw_idx, inner_idx, A_in, A_hw = unit_split(indptr, indices, n)
    np.testing.assert_array_equal(d[:, hw_idx], conv[:, hw_idx])
    assert d[:, inner_idx].min() >= 0.0
    c1i, c2i, c3i = c1[inner_idx], c2[inner_idx], c3[inner_idx]
    uh = oracle.UnitHydrograph(kern)
    conv_ref = uh.convolve(depth[:3])
    qc, qf, d_ref = np.zeros(inner_idx.size), np.zeros(inner_idx.size), np.zeros((3, n))
    oracle.unit_route(A_in.indptr, A_in.indices, -c1i[A_in.indices], A_in.indptr, A_in.indices, A_in.data,
                      A_hw.indptr, A_hw.indices, A_hw.data, c1i, c2i, c3i, hw_idx, inner_idx, qc, qf, conv_ref, d_ref, 1)
    assert_close(conv[:3], conv_ref, 'convolved')
    assert_close(d[:3], d_ref, 'discharge')

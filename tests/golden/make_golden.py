"""
Generates tests/golden/*.npz by RUNNING THE REFERENCE (read-only at /root/reference) on small seeded inputs.

Runs only in the build container: it refuses to start where /root/reference is absent (the GPU box), and the
reference's source never enters this repository -- only inputs and the outputs it produced are written.

How the reference is driven: its hot-path modules are imported from where they lie.
numba, netCDF4, xarray, geopandas and shapely are not installed here, so `numba.njit` is replaced by the
identity decorator (the three kernels then execute as interpreted Python, statement for statement;
`fastmath` has no meaning without the JIT) and the I/O-only packages are empty placeholder modules that
no code on the path touches.  typing.Self (Python 3.12) is aliased from typing_extensions.

    python tests/golden/make_golden.py
"""
import importlib
import os
import sys
import tempfile
import types
import typing

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def load_reference():
    if not os.path.isdir(os.path.join(REF, 'river_route')):
        raise SystemExit('make_golden.py: /root/reference is not present; golden vectors can only be '
                         'regenerated in the build container')
    import typing_extensions
    if not hasattr(typing, 'Self'):
        typing.Self = typing_extensions.Self
    numba = types.ModuleType('numba')
    numba.njit = lambda *a, **k: (a[0] if a and callable(a[0]) and not k else (lambda f: f))
    sys.modules['numba'] = numba
    for name in ('netCDF4', 'xarray', 'geopandas', 'shapely', 'shapely.geometry', 'shapely.ops'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['xarray'].Dataset = object
    sys.modules['geopandas'].GeoDataFrame = object
    shell = types.ModuleType('river_route')
    shell.__path__ = [os.path.join(REF, 'river_route')]
    sys.modules['river_route'] = shell
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only reference
    routers = importlib.import_module('river_route.routers')
    kernels = importlib.import_module('river_route.routers._numba_kernels')
    uhk = importlib.import_module('river_route.uhkernels.UnitHydrograph')
    tools = importlib.import_module('river_route.tools')
    return routers, kernels, uhk, tools


def main():
    routers, K, uhk, tools = load_reference()
    sys.path.insert(0, REPO)
    from river_route_amd import synth  # seeded inputs only
    import pandas as pd
    import scipy.sparse

    rng = np.random.default_rng(20260320)
    out: dict[str, np.ndarray] = {}

    def ref_coeffs(k, x, A, dt):
        fake = types.SimpleNamespace(k=k, x=x, A=A, logger=types.SimpleNamespace(
            debug=lambda *a: None, warning=lambda *a: None))
        routers.Muskingum._set_muskingum_coefficients(fake, dt)
        return fake

    def network_case(tag, river_ids, downstream_ids, k, x, dt, T, nsubs, n_ks_list):
        n = len(river_ids)
        A = tools.adjacency_matrix(river_ids, downstream_ids)
        f = ref_coeffs(k, x, A, dt)
        out[f'{tag}/river_ids'] = river_ids
        out[f'{tag}/downstream_ids'] = downstream_ids
        out[f'{tag}/k'] = k
        out[f'{tag}/x'] = x
        out[f'{tag}/dt'] = np.float64(dt)
        out[f'{tag}/indptr'] = f._csc_indptr
        out[f'{tag}/indices'] = f._csc_indices
        for nm in ('c1', 'c2', 'c3'):
            out[f'{tag}/{nm}'] = getattr(f, nm)
        out[f'{tag}/lhs_off'] = f._lhs_off_data
        q0 = rng.uniform(0.0, 5.0, n)
        ql = rng.uniform(0.0, dt, (T, n))
        ql[rng.uniform(size=ql.shape) < 0.1] = 0.0
        out[f'{tag}/q0'] = q0
        out[f'{tag}/qlateral'] = ql
        for nsub in nsubs:
            # rapid (RapidMuskingum.py:24-32): c4_dt = (c1+c2)/dt_runoff with dt_runoff = nsub*dt
            c4_dt = (f.c1 + f.c2) / (dt * nsub)
            q_t = q0.copy()
            d = np.zeros((T, n))
            K.rapid_route(f._csc_indptr, f._csc_indices, f._lhs_off_data, f.c2, f.c3, c4_dt, q_t, ql, d, nsub)
            out[f'{tag}/rapid{nsub}/c4_dt'] = c4_dt
            out[f'{tag}/rapid{nsub}/q_t'] = q_t
            out[f'{tag}/rapid{nsub}/discharge'] = d
            # channel-only (Muskingum.py:276-286)
            n_out = max(T // 4, 1)
            q_t = q0.copy()
            d = np.zeros((n_out, n))
            K.muskingum_route(f._csc_indptr, f._csc_indices, f._lhs_off_data, f.c2, f.c3, q_t, d, n_out, nsub)
            out[f'{tag}/musk{nsub}/q_t'] = q_t
            out[f'{tag}/musk{nsub}/discharge'] = d
        # zero state => exactly zero output (tests/test_muskingum.py:48-71)
        q_t = np.zeros(n)
        d = np.zeros((3, n))
        K.muskingum_route(f._csc_indptr, f._csc_indices, f._lhs_off_data, f.c2, f.c3, q_t, d, 3, 2)
        assert not d.any() and not q_t.any()

        # unit (UnitMuskingum.py:40-98), driven exactly as _hook_before_route/_router do
        incoming = np.asarray(A.sum(axis=1)).flatten()
        hw_idx = np.where(incoming == 0)[0]
        inner_idx = np.where(incoming != 0)[0]
        A_in = A[np.ix_(inner_idx, inner_idx)].tocsc()
        A_hw = A[np.ix_(inner_idx, hw_idx)].tocsc()
        c1i, c2i, c3i = f.c1[inner_idx], f.c2[inner_idx], f.c3[inner_idx]
        lhs_in = np.ascontiguousarray(-c1i[A_in.indices])
        out[f'{tag}/hw_idx'] = hw_idx
        out[f'{tag}/inner_idx'] = inner_idx
        for n_ks in n_ks_list:
            kern = rng.uniform(0.0, 1.0, (n_ks, n)) * rng.uniform(10.0, 100.0, n)[None, :]
            state0 = rng.uniform(0.0, 2.0, (n_ks, n))
            depth = rng.uniform(0.0, 0.2, (T, n))
            with tempfile.TemporaryDirectory() as tmp:
                kp = os.path.join(tmp, 'k.npz')
                scipy.sparse.save_npz(kp, scipy.sparse.csr_matrix(kern))
                uh = uhk.UnitHydrograph(kp)
            uh.state = state0.copy()
            conv = np.ascontiguousarray(uh.convolve(depth))
            out[f'{tag}/unit_ks{n_ks}/kernel'] = kern
            out[f'{tag}/unit_ks{n_ks}/state0'] = state0
            out[f'{tag}/unit_ks{n_ks}/depth'] = depth
            out[f'{tag}/unit_ks{n_ks}/convolved'] = conv
            out[f'{tag}/unit_ks{n_ks}/state1'] = uh.state.copy()
            for nsub in nsubs:
                q_ch = q0[inner_idx].copy()
                q_full = q_ch.copy()
                d = np.zeros((T, n))
                K.unit_route(A_in.indptr, A_in.indices, lhs_in,
                             A_in.indptr, A_in.indices, np.ascontiguousarray(A_in.data),
                             A_hw.indptr, A_hw.indices, np.ascontiguousarray(A_hw.data),
                             c1i, c2i, c3i, hw_idx, inner_idx, q_ch, q_full, conv, d, nsub)
                out[f'{tag}/unit_ks{n_ks}/nsub{nsub}/q_ch'] = q_ch
                out[f'{tag}/unit_ks{n_ks}/nsub{nsub}/q_full'] = q_full
                out[f'{tag}/unit_ks{n_ks}/nsub{nsub}/discharge'] = d

    # (i) the 9-reach network of docs/references/math.md:70-80
    rid = np.arange(1, 10, dtype=np.int64)
    did = np.array([5, 5, 6, 6, 7, 7, 9, 9, -1], dtype=np.int64)
    network_case('docs9', rid, did, rng.uniform(900, 7200, 9), rng.uniform(0.05, 0.45, 9), 900.0, 12, (1, 4), (1, 3))

    # (ii) ~1k-reach seeded synthetic tree (BASELINE config 1 stand-in), two outlets via a second small tree
    net = synth.synth_network(1000)
    network_case('tree1k', net.river_ids, net.downstream_ids, net.k, net.x, 900.0, 40, (1, 4), (3,))

    # (iii) a forest with chains, 1-reach components and in-degree 3
    rid = np.arange(100, 130, dtype=np.int64)
    did = np.full(30, -1, dtype=np.int64)
    for up, dn in [(0, 5), (1, 5), (2, 5), (5, 6), (6, 7), (7, 8), (8, 20), (3, 9), (9, 20), (10, 11), (11, 12),
                   (12, 13), (13, 14), (14, 29), (20, 29), (15, 16), (21, 22), (22, 23), (24, 25)]:
        did[up] = rid[dn]
    network_case('forest30', rid, did, rng.uniform(300, 9000, 30), rng.uniform(0.0, 0.5, 30), 600.0, 60, (1, 3), (48,))

    # (iv) UH convolution alone, incl. T < n_ks and carried state across two calls (UnitHydrograph.py:77-107)
    for ci, (T, n_ks) in enumerate([(10, 3), (2, 6), (1, 4), (7, 1), (60, 48)]):
        nb = 5
        kern = rng.uniform(0.0, 1.0, (n_ks, nb))
        lat_a = rng.uniform(0.0, 1.0, (T, nb))
        lat_b = rng.uniform(0.0, 1.0, (T + 1, nb))
        st0 = rng.uniform(0.0, 1.0, (n_ks, nb))
        with tempfile.TemporaryDirectory() as tmp:
            kp = os.path.join(tmp, 'k.npz')
            scipy.sparse.save_npz(kp, scipy.sparse.csr_matrix(kern))
            uh = uhk.UnitHydrograph(kp)
            uh2 = uhk.UnitHydrograph(kp)
        uh.state = st0.copy()
        ra = uh.convolve(lat_a).copy()
        sa = uh.state.copy()
        rb = uh.convolve(lat_b).copy()
        sb = uh.state.copy()
        inc = np.stack([uh2.convolve_incrementally(r) for r in np.vstack([lat_a, lat_b])])
        for nm, v in dict(kernel=kern, lat_a=lat_a, lat_b=lat_b, state0=st0, out_a=ra, state_a=sa,
                          out_b=rb, state_b=sb, incremental_zero_state=inc).items():
            out[f'conv{ci}/{nm}'] = v

    # (v) adjacency rejections restated from tests/test_tools.py:48-60 are known-answer tests in test_oracle.py;
    #     (vi) coefficient failure: k = 0 -> ValueError
    try:
        ref_coeffs(np.array([3600.0, 0.0]), np.array([0.2, 0.2]), scipy.sparse.csc_matrix((2, 2)), 900.0)
        raise AssertionError('expected ValueError')
    except ValueError as e:
        out['coeff_fail/message'] = np.array(str(e))

    # (vii) SCS kernel builders (uhkernels/_SCSBase.py:36-76) -- what produces a realistic uh_kernel_file
    scs_mod = importlib.import_module('river_route.uhkernels')
    tc = np.array([7200.0, 14400.0, 3600.0, 900.0, 40000.0])
    area = np.array([1e6, 5e6, 2e5, 3.3e7, 1.0e8])
    for tr in (3600.0, 900.0):
        out[f'scs/tr{int(tr)}/triangular'] = scs_mod.SCSTriangular(tr=tr, tc=tc, area=area).kernel
        out[f'scs/tr{int(tr)}/curvilinear'] = scs_mod.SCSCurvilinear(tr=tr, tc=tc, area=area).kernel
    out['scs/tc'], out['scs/area'] = tc, area

    np.savez_compressed(os.path.join(HERE, 'kernels.npz'), **out)
    print('wrote kernels.npz with', len(out), 'arrays')

    # ---- router level: RapidMuskingum / UnitMuskingum / Muskingum .route() with in-memory inputs ----
    rout: dict[str, np.ndarray] = {}
    net = synth.synth_network(300, seed=7)
    n = net.n
    with tempfile.TemporaryDirectory() as tmp:
        params = os.path.join(tmp, 'params.parquet')
        pd.DataFrame({'river_id': net.river_ids, 'downstream_river_id': net.downstream_ids,
                      'k': net.k, 'x': net.x}).to_parquet(params)
        init = os.path.join(tmp, 'init.parquet')
        q0 = rng.uniform(0.0, 5.0, n)
        pd.DataFrame({'Q': q0}).to_parquet(init)
        files = []
        for i in range(2):
            p = os.path.join(tmp, f'ql{i}.nc')
            open(p, 'w').close()
            files.append(p)
        T = 24
        dates = [np.datetime64('2020-01-01T00:00:00') + np.arange(i * T, (i + 1) * T) * np.timedelta64(3600, 's')
                 for i in range(2)]
        dates = [d.astype('datetime64[s]') for d in dates]
        vols = [rng.uniform(0.0, 3600.0, (T, n)) for _ in range(2)]
        depths = [rng.uniform(0.0, 0.2, (T, n)).astype(np.float32).astype(np.float64) for _ in range(2)]
        rout['river_ids'], rout['downstream_ids'], rout['k'], rout['x'], rout['q0'] = \
            net.river_ids, net.downstream_ids, net.k, net.x, q0
        for i in range(2):
            rout[f'vol{i}'], rout[f'depth{i}'], rout[f'dates{i}'] = vols[i], depths[i], dates[i].astype(np.int64)

        def drive(cls, tag, series, **cfg):
            class InMem(cls):
                def _qlateral_generator(self):
                    for d, a, fin, fout in zip(dates, series, self.cfg.qlateral_files, self.cfg.discharge_files):
                        yield d, a, fin, fout
            got = []
            r = InMem(params_file=params, qlateral_files=files, discharge_dir=tmp, log=False, **cfg)
            r.set_write_discharges(lambda d, q, f, rf='': got.append((np.asarray(d), np.asarray(q))))
            r.route()
            for i, (d, q) in enumerate(got):
                rout[f'{tag}/dates{i}'] = d.astype('datetime64[s]').astype(np.int64)
                rout[f'{tag}/q{i}'] = q
                assert q.dtype == np.float32
            rout[f'{tag}/final_state'] = r.channel_state.copy()
            return r

        drive(routers.RapidMuskingum, 'rapid_seq', vols, channel_state_init_file=init, dt_routing=900)
        drive(routers.RapidMuskingum, 'rapid_seq_resample', vols, channel_state_init_file=init,
              dt_routing=1800, dt_discharge=3 * 3600)
        drive(routers.RapidMuskingum, 'rapid_ens', vols, channel_state_init_file=init,
              runoff_processing_mode='ensemble')

        n_ks = 5
        kern = rng.uniform(0.0, 1.0, (n_ks, n)) * rng.uniform(10.0, 100.0, n)[None, :]
        kp = os.path.join(tmp, 'uh.npz')
        scipy.sparse.save_npz(kp, scipy.sparse.csr_matrix(kern))
        uh0 = rng.uniform(0.0, 2.0, (n_ks, n))
        uhs = os.path.join(tmp, 'uhstate.parquet')
        pd.DataFrame(uh0.T).to_parquet(uhs)
        rout['uh_kernel'], rout['uh_state0'] = kern, uh0
        r = drive(routers.UnitMuskingum, 'unit_seq', depths, channel_state_init_file=init, dt_routing=1200,
                  uh_kernel_file=kp, uh_state_init_file=uhs)
        rout['unit_seq/uh_state_final'] = r._uh.state.copy()
        r = drive(routers.UnitMuskingum, 'unit_ens', depths, channel_state_init_file=init,
                  uh_kernel_file=kp, runoff_processing_mode='ensemble')
        rout['unit_ens/uh_state_final'] = r._uh.state.copy()

        got = []
        m = routers.Muskingum(params_file=params, discharge_files=[os.path.join(tmp, 'd.nc')], log=False,
                              channel_state_init_file=init, dt_routing=900, dt_total=6 * 3600, dt_discharge=1800,
                              start_datetime='2021-03-04')
        m.set_write_discharges(lambda d, q, f, rf='': got.append((np.asarray(d), np.asarray(q))))
        m.route()
        rout['musk/dates'] = got[0][0].astype('datetime64[s]').astype(np.int64)
        rout['musk/q'] = got[0][1]
        rout['musk/final_state'] = m.channel_state.copy()

    np.savez_compressed(os.path.join(HERE, 'routers.npz'), **rout)
    print('wrote routers.npz with', len(rout), 'arrays')


if __name__ == '__main__':
    main()

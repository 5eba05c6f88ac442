import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
from oracle import oracle
from river_route_amd import synth
from river_route_amd.engine import Plan, DeviceBuffer
def csc(down):
    has = down >= 0
    return np.concatenate([[0], np.cumsum(has)]).astype(np.int32), down[has].astype(np.int32)
def run(n, T, nsub, env):
    for k in ('RR_WAVE','RR_WAVE_K','RR_WAVE_PPT','RR_TILE_BLOCK'): os.environ.pop(k, None)
    os.environ.update(env)
    net = synth.synth_network(n, seed=13)
    indptr, indices = csc(net.down_index)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0)
    lhs, c4 = -c1[indices], (c1 + c2) / (900.0 * nsub)
    ql = synth.synth_qlateral(n, 0, T, dt=900.0*nsub); q0 = 4.0 * synth.u01(1, np.arange(n))
    qr, dr = q0.copy(), np.zeros((T, n))
    oracle.rapid_route(indptr, indices, lhs, c2, c3, c4, qr, ql, dr, nsub)
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(lhs, c2, c3, c4)
        dq = DeviceBuffer(n*8).upload(q0); dl = DeviceBuffer(ql.nbytes).upload(ql); do = DeviceBuffer(T*n*8)
        plan.rapid_route_dev(dq, dl, T, do, T, T, nsub)
        q = dq.download(np.float64, (n,)); d = do.download(np.float64, (T, n))
    scale = np.abs(dr).max()
    err = np.abs(d - dr).max() / scale; errq = np.abs(q - qr).max() / scale
    bad = np.argwhere(np.abs(d - dr) > 1e-9 * scale)
    print(f'n={n} T={T} nsub={nsub} env={env}: rel err d={err:.2e} q={errq:.2e} bad={len(bad)} first={bad[:3].tolist()}', flush=True)
os.environ['RR_VERBOSE'] = '1'
run(3000, 40, 1, {'RR_WAVE': '1'})
run(3000, 40, 1, {'RR_WAVE': '1', 'RR_TILE_BLOCK': '64'})
run(60000, 70, 1, {'RR_WAVE': '1', 'RR_TILE_BLOCK': '64', 'RR_WAVE_K': '32'})
run(60000, 70, 1, {'RR_WAVE': '1', 'RR_WAVE_K': '64'})
run(60000, 33, 3, {'RR_WAVE': '1', 'RR_WAVE_K': '32', 'RR_TILE_BLOCK': '333'})
run(300000, 70, 1, {'RR_WAVE': '1'})
run(1000000, 80, 1, {})


def run_unit_fused(n, T, n_ks):
    from conftest import unit_split
    for k in ('RR_WAVE','RR_WAVE_K','RR_WAVE_THREADS','RR_TILE_BLOCK'): os.environ.pop(k, None)
    net = synth.synth_network(n, seed=23)
    indptr, indices = csc(net.down_index)
    hw_idx, inner_idx, A_in, A_hw = unit_split(indptr, indices, n)
    c1, c2, c3 = oracle.muskingum_coefficients(net.k, net.x, 900.0)
    c1i, c2i, c3i = c1[inner_idx], c2[inner_idx], c3[inner_idx]
    kern = synth.synth_uh_kernel(n, n_ks)
    uh = oracle.UnitHydrograph(kern)
    depth = synth.synth_runoff_depth(n, 0, T)
    conv = uh.convolve(depth)
    q0 = 3.0 * synth.u01(7, np.arange(n))
    qc, qf, dr = q0[inner_idx].copy(), q0[inner_idx].copy(), np.zeros((T, n))
    oracle.unit_route(A_in.indptr, A_in.indices, -c1i[A_in.indices], A_in.indptr, A_in.indices, A_in.data, A_hw.indptr, A_hw.indices, A_hw.data,
                      c1i, c2i, c3i, hw_idx, inner_idx, qc, qf, conv, dr, 1)
    ni = inner_idx.size
    with Plan(indptr, indices) as plan:
        plan.set_coeffs(-c1[indices], c2, c3, None)
        bufs = dict(k=DeviceBuffer(kern.nbytes).upload(kern), s=DeviceBuffer(kern.nbytes).upload(np.zeros_like(kern)), d=DeviceBuffer(depth.nbytes).upload(depth),
                    o=DeviceBuffer(T * n * 8), f=DeviceBuffer(n * 8), qc=DeviceBuffer(ni * 8).upload(q0[inner_idx].copy()), qf=DeviceBuffer(ni * 8).upload(q0[inner_idx].copy()))
        plan.unit_route_uh_dev(bufs['qc'], bufs['qf'], bufs['f'], bufs['k'], bufs['s'], n_ks, bufs['d'], T, 1, discharge=bufs['o'])
        d = bufs['o'].download(np.float64, (T, n)); st = bufs['s'].download(np.float64, kern.shape)
    scale = np.abs(dr).max()
    print(f'unit fused n={n} T={T} n_ks={n_ks}: rel err d={np.abs(d - dr).max() / scale:.2e} uh state={np.abs(st - uh.state).max() / scale:.2e} bad={int((np.abs(d - dr) > 1e-9 * scale).sum())}', flush=True)

run_unit_fused(60000, 200, 48)
run_unit_fused(60000, 150, 12)

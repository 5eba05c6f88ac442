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

"""rr_runoff_to_qlateral_dev at routing scale: 1M catchments, a 500 x 500-cell grid region,
~3 cells per catchment, one month of hourly float32 runoff (744 steps), device-resident arrays, HIP-event timing on
the stream the kernel is launched on.  Algorithmic bytes = the (T, n) float64 result written once + each stored
weight's T float32 values gathered once + the CSR once."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from river_route_amd import _lib

n, npts, T = 1_000_000, 250_000, 744
rng = np.random.default_rng(0)
cnt = rng.integers(1, 6, n)
indptr = np.zeros(n + 1, np.int32); indptr[1:] = np.cumsum(cnt)
nnz = int(indptr[-1])
base = rng.integers(0, npts, n)                       # a catchment's cells are neighbours on the grid
indices = ((np.repeat(base, cnt) + rng.integers(0, 3, nnz) + 500 * rng.integers(0, 3, nnz)) % npts).astype(np.int32)
weights = rng.random(nnz)
dev = torch.device('cuda:0')
d_ip, d_ix, d_w = torch.from_numpy(indptr).to(dev), torch.from_numpy(indices).to(dev), torch.from_numpy(weights).to(dev)
area = torch.rand(n, dtype=torch.float64, device=dev) * 1e7
for layout in ('point-major padded', 'point-major', 'time-major'):
    tp = -(-T // 16) * 16 if layout == 'point-major padded' else T
    runoff = torch.rand((npts, tp) if layout != 'time-major' else (T, npts), dtype=torch.float32, device=dev)
    st, sp = (1, tp) if layout != 'time-major' else (npts, 1)
    out = torch.empty((T, n), dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ms = []
    for rep in range(4):
        ev[0].record(s)
        rc = _lib.lib().rr_runoff_to_qlateral_dev(0, n, npts, T, d_ip.data_ptr(), d_ix.data_ptr(), d_w.data_ptr(), runoff.data_ptr(), 1,
                                                  st, sp, area.data_ptr(), 2, out.data_ptr(), s.cuda_stream)
        assert rc == 0
        ev[1].record(s); torch.cuda.synchronize()
        ms.append(ev[0].elapsed_time(ev[1]))
    alg = T * n * 8 + nnz * T * 4 + nnz * 12 + n * 12
    print(f'{layout}: {min(ms):.2f} ms for {n} catchments x {T} steps ({nnz} weights) -> {alg / min(ms) / 1e6:.0f} GB/s algorithmic, '
          f'{n * T / (min(ms) * 1e-3):.3g} catchment-steps/s')

# ---- the routers' grid-runoff path end to end (one month at 1M reaches): two calls vs the fused one ----
from river_route_amd import synth
from river_route_amd.engine import Plan
net = synth.synth_network(n)
has = net.down_index >= 0
cptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32); cidx = net.down_index[has].astype(np.int32)
r = 900.0 / net.k; den = r + 2 * (1 - net.x)
c1, c2, c3 = (r - 2 * net.x) / den, (r + 2 * net.x) / den, (2 * (1 - net.x) - r) / den
plan = Plan(cptr, cidx); plan.set_coeffs(-c1[cidx], c2, c3, (c1 + c2) / 900.0)
tp = -(-T // 16) * 16
runoff = torch.rand((npts, tp), dtype=torch.float32, device=dev) * 1e-3
q = torch.zeros(n, dtype=torch.float64, device=dev)
ql = torch.empty((T, n), dtype=torch.float64, device=dev)
out32 = torch.empty((T, n), dtype=torch.float32, device=dev)
s = torch.cuda.current_stream()
def two_calls():
    rc = _lib.lib().rr_runoff_to_qlateral_dev(0, n, npts, T, d_ip.data_ptr(), d_ix.data_ptr(), d_w.data_ptr(), runoff.data_ptr(), 1, 1, tp,
                                              area.data_ptr(), 2, ql.data_ptr(), s.cuda_stream)
    assert rc == 0
    plan.rapid_route_f32_dev(q, ql, T, out32, T, 1, 1, s.cuda_stream)
def fused():
    plan.rapid_route_runoff_dev(q, npts, d_ip, d_ix, d_w, runoff, True, 1, tp, area, 2, T, discharge32=out32, stream=s.cuda_stream)
for name, fn in (('runoff_to_qlateral_dev + rapid_route_f32_dev', two_calls), ('rapid_route_runoff_dev (fused)', fused)):
    ms = []
    for rep in range(4):
        q.zero_(); ev[0].record(s); fn(); ev[1].record(s); torch.cuda.synchronize(); ms.append(ev[0].elapsed_time(ev[1]))
    print(f'{name}: {min(ms):.2f} ms for {n} reaches x {T} hourly steps, float32 discharge out')

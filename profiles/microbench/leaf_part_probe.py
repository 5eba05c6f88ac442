"""Why does a leaf part of the 10M / 8 cut take longer than 1.25 x the 1M bench?  The same local network routed (a) as a part
(boundary reaches, streaming session), (b) as a plain plan through rr_rapid_route_dev, (c) a connected 1.25M-reach network."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, '.')
from river_route_amd import synth
from river_route_amd.engine import Plan, partition_forest
from river_route_amd.multi_gpu import split_network, HipPartEngine

world, per = 8, 1_250_000
cases = sys.argv[1] if len(sys.argv) > 1 else 'abc'
shift_mb = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # a dummy allocation made first, to move every later buffer
sink = int(sys.argv[3]) if len(sys.argv) > 3 else 96          # rows of the cyclic discharge sink
n, T, dt = per * world, 35040, 900.0
net = synth.synth_network(n, order='random')
has = net.down_index >= 0
indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
indices = net.down_index[has].astype(np.int32)
part_of, sizes = partition_forest(indptr, indices, world)
r = dt / net.k
den = r + 2.0 * (1.0 - net.x)
c1, c2, c3 = (r - 2.0 * net.x) / den, (r + 2.0 * net.x) / den, (2.0 * (1.0 - net.x) - r) / den
spec = split_network(net.down_index, part_of, 0, world)
idx = (np.arange(96, dtype=np.uint64)[:, None] * np.uint64(n)) + spec.real_global.astype(np.uint64)[None, :]
lateral = dt * synth.u01(synth.FORCING_SEED, idx)
dev = torch.device('cuda:0')
dummy = torch.empty(shift_mb << 20, dtype=torch.uint8, device=dev) if shift_mb else None

def timed(fn, label):
    ts = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(label, ' '.join(f'{t * 1e3:.1f}' for t in ts), 'ms', flush=True)

if 'a' in cases:
    eng = HipPartEngine(spec, c1, c2, c3, (c1 + c2) / dt, np.zeros(n), lateral, T, 1, 0)
    timed(lambda: (eng.begin(), eng.advance(T, T), eng.end()), f'(a) part 0 as a part ({spec.real_global.size} reaches, {spec.export_global.size} exports):')
    del eng; torch.cuda.empty_cache()

members = spec.real_global
hasl = spec.down_local >= 0
s = torch.cuda.current_stream().cuda_stream
with Plan(spec.indptr, spec.indices) as plan:
    plan.set_coeffs(-c1[members][spec.down_local[hasl]], c2[members], c3[members], ((c1 + c2) / dt)[members])
    ql = torch.from_numpy(lateral).to(dev); out = torch.zeros((sink, members.size), dtype=torch.float64, device=dev)
    q = torch.zeros(members.size, dtype=torch.float64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    if 'b' in cases: timed(lambda: plan.rapid_route_dev(q, ql, 96, out, sink, T, 1, s), '(b) the same local network as a plain plan:')
    print('   ', plan.tile_info(), 'depth', plan.depth)
del ql, out; torch.cuda.empty_cache()

net2 = synth.synth_network(per, order='random')
has2 = net2.down_index >= 0
ip2 = np.concatenate([[0], np.cumsum(has2)]).astype(np.int32); ix2 = net2.down_index[has2].astype(np.int32)
r2 = dt / net2.k; den2 = r2 + 2.0 * (1.0 - net2.x)
d1, d2, d3 = (r2 - 2.0 * net2.x) / den2, (r2 + 2.0 * net2.x) / den2, (2.0 * (1.0 - net2.x) - r2) / den2
with Plan(ip2, ix2) as plan:
    plan.set_coeffs(-d1[ix2], d2, d3, (d1 + d2) / dt)
    ql = torch.from_numpy(synth.synth_qlateral(per, 0, 96)).to(dev); out = torch.zeros((sink, per), dtype=torch.float64, device=dev)
    q = torch.zeros(per, dtype=torch.float64, device=dev)
    if 'c' in cases: timed(lambda: plan.rapid_route_dev(q, ql, 96, out, sink, T, 1, s), '(c) one connected 1.25M-reach network:')
    print('   ', plan.tile_info(), 'depth', plan.depth)

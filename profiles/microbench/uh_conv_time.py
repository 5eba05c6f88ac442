"""Wall time of rr_uh_convolve_dev at BASELINE config 4's shape (1M reaches x 3,504 rows x 48 taps; the call
synchronises, so perf_counter brackets it; includes the carry-over tail kernel and state copy, ~3 ms)."""
import sys, time
import torch
sys.path.insert(0, '.')
from river_route_amd.engine import uh_convolve_dev

n, T, n_ks = 1_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 3504, int(sys.argv[2]) if len(sys.argv) > 2 else 48
dev = torch.device('cuda:0')
kern = torch.rand(n_ks, n, dtype=torch.float64, device=dev); kern /= kern.sum(0, keepdim=True)
state = torch.zeros(n_ks, n, dtype=torch.float64, device=dev)
lat = torch.rand(T, n, dtype=torch.float64, device=dev)
out = torch.empty_like(lat)
torch.cuda.synchronize()
s = torch.cuda.current_stream().cuda_stream
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    uh_convolve_dev(kern, state, lat, out, T, n_ks, n, device=0, stream=s)
    ts.append((time.perf_counter() - t0) * 1e3)
print(f'T={T} n_ks={n_ks}: ' + ' '.join(f'{t:.2f}' for t in ts) + f' ms  -> {2 * 8.0 * n * T / (min(ts) * 1e-3) / 1e12:.2f} TB/s')

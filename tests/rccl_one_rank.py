"""RCCL on a one-GPU box, run as a child process by tests/test_multi.py::test_rccl_call_sequence_on_one_rank: a world of one
rank makes every RCCL call bench.py's N > 1 leg makes -- init with a device and a timeout, barrier, all_reduce (MAX and MIN),
all_gather, and the exchange's own point-to-point calls as far as one rank can make them."""
import datetime
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from river_route_amd import multi_gpu  # noqa: E402,F401  (sets HSA_ENABLE_IPC_MODE_LEGACY=0 before the HIP runtime starts)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('RANK', '0')
os.environ.setdefault('WORLD_SIZE', '1')
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', device_id=dev, timeout=datetime.timedelta(seconds=120))
assert dist.get_backend() == 'nccl'
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
f = torch.tensor([1.0], dtype=torch.float64, device=dev)
dist.all_reduce(f, op=dist.ReduceOp.MIN)
g = [torch.zeros(3, dtype=torch.float64, device=dev)]
dist.all_gather(g, torch.arange(3, dtype=torch.float64, device=dev))
# The exchange's calls (multi_gpu.run_distributed): receives posted as a batch into staging buffers, a send as a batch of one,
# completion polled through Work.is_completed(), the stream-side wait, the copy into a column slice of the boundary series.
# Two ranks post the receive and the send in separate batches; a rank talking to itself has to group them (RCCL matches a
# send and a receive of one rank only inside one group call), so here they share a batch.
series = torch.zeros((128, 7), dtype=torch.float64, device=dev)
src = torch.arange(128 * 3, dtype=torch.float64, device=dev).reshape(128, 3)
buf = series[:, 2:5].new_empty((128, 3))
works = dist.batch_isend_irecv([dist.P2POp(dist.irecv, buf, 0), dist.P2POp(dist.isend, src.contiguous(), 0)])
for w in works:
    w.wait()
series[:, 2:5].copy_(buf)
torch.cuda.synchronize()
assert all(w.is_completed() for w in works)
assert torch.equal(series[:, 2:5], src) and float(series[:, :2].abs().sum()) == 0.0
assert float(t.item()) == 1.5 and float(f.item()) == 1.0 and g[0].tolist() == [0.0, 1.0, 2.0]
print('rccl ok', dist.get_backend(), flush=True)
dist.destroy_process_group()

"""RCCL on a one-GPU box: world of one rank -- init, barrier, all_reduce, all_gather, the calls bench.py's N > 1 leg makes
besides the point-to-point exchange (which needs a second GPU)."""
import os
import torch
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
dist.init_process_group('nccl', device_id=dev)
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
g = [torch.zeros(3, dtype=torch.float64, device=dev)]
dist.all_gather(g, torch.arange(3, dtype=torch.float64, device=dev))
# the exchange's own calls, as far as one rank can make them: a batched receive and a batched send of float64 rows (to itself, so
# both in one batch -- two ranks post them apart, see multi_gpu.run_distributed), the wait on the work, the copy into a column slice
series = torch.zeros((128, 7), dtype=torch.float64, device=dev)
src = torch.arange(128 * 3, dtype=torch.float64, device=dev).reshape(128, 3)
buf = series[:, 2:5].new_empty((128, 3))
works = dist.batch_isend_irecv([dist.P2POp(dist.irecv, buf, 0), dist.P2POp(dist.isend, src.contiguous(), 0)])
for w in works:
    w.wait()
series[:, 2:5].copy_(buf)
torch.cuda.synchronize()
assert torch.equal(series[:, 2:5], src) and float(series[:, :2].abs().sum()) == 0.0
print('rccl ok', float(t.item()), g[0].tolist(), dist.get_backend())
dist.destroy_process_group()

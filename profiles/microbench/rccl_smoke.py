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
torch.cuda.synchronize()
print('rccl ok', float(t.item()), g[0].tolist(), dist.get_backend())
dist.destroy_process_group()

"""Developer check (GPU box): RCCL comes up in this image the way bench.py / Trainer use it (1 rank, 1 GPU)."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.arange(8, device=dev, dtype=torch.float32)
dist.all_reduce(t)
dist.broadcast(t, src=0)
dist.barrier()
x = torch.tensor([1.5], device=dev, dtype=torch.float64)
dist.all_reduce(x, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
print("nccl ok", t.tolist(), float(x))
dist.destroy_process_group()

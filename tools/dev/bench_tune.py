"""Developer timing of one fitting step with the generator being tuned (GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_train import Args
from hfa_gp_amd.trainer import Trainer
from hfa_gp_amd.synthetic import look_at_label

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0"); torch.manual_seed(0)
tr = Trainer(Args(), dev, mode="3dmm", lpips="none"); tr.tune_generator()
g = torch.Generator().manual_seed(1)
real = (0.5 * torch.randn(B, 3, 256, 256, generator=g)).clamp(-1, 1).to(dev); params = torch.randn(B, 76, generator=g).to(dev)
lab = look_at_label(1.57 + 0.3 * torch.randn(B, generator=g), 1.57 + 0.15 * torch.randn(B, generator=g), flipped=False).to(dev)
for _ in range(2):
    tr.gen_update(real, lab.clone(), params)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(iters):
    l2 = tr.gen_update(real, lab.clone(), params)[1]
torch.cuda.synchronize()
print("tune step B=%d: %.2f ms/step, l2 %.4f, mem %.1f GiB" % (B, (time.perf_counter() - t) / iters * 1e3, float(l2), torch.cuda.max_memory_allocated() / 2**30))

"""Developer report (GPU box): which source lines launch the framework's small kernels in one fitting step?
usage: gpu_dev_opsites.py [rgb|3dmm] [tuned]"""
import os
import sys
import collections

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_train import Args  # noqa: E402
from hfa_gp_amd.trainer import Trainer  # noqa: E402
from hfa_gp_amd.synthetic import look_at_label  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "3dmm"
B = 2
dev = torch.device("cuda:0")
torch.manual_seed(0)
tr = Trainer(Args(), dev, mode=mode, lpips="none")
if len(sys.argv) > 2 and sys.argv[2] == "tuned":
    tr.tune_generator()
g = torch.Generator().manual_seed(1)
real = (0.5 * torch.randn(B, 3, 256, 256, generator=g)).clamp(-1, 1).to(dev)
params = torch.randn(B, 76, generator=g).to(dev)
label0 = look_at_label(1.57 + 0.3 * torch.randn(B, generator=g), 1.57 + 0.15 * torch.randn(B, generator=g), flipped=False).to(dev)
step = (lambda: tr.gen_update(real, label0.clone())) if mode == "rgb" else (lambda: tr.gen_update(real, label0.clone(), params))
for _ in range(4):
    step()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

sites = collections.Counter()
want = ("mul", "copy_", "fill_", "add_", "add", "sum", "zero_", "clone", "zeros", "zeros_like", "abs", "lt", "mul_", "empty_like",
        "_foreach_add_", "neg", "sub", "div", "sqrt", "rsqrt", "mean", "cat", "stack", "index", "where", "clamp", "expand", "_to_copy")


class Sites(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in want:
            st = [f for f in traceback.extract_stack() if "hfa" in f.filename and "tools/dev" not in f.filename]
            w = st[-1] if st else None
            sites[(name, f"{os.path.basename(w.filename)}:{w.lineno} {w.line[:90]}" if w else "?")] += 1
        return func(*args, **(kwargs or {}))


with Sites():
    step()
torch.cuda.synchronize()
for (name, where), c in sorted(sites.items(), key=lambda kv: (-kv[1], kv[0][1]))[:90]:
    print(f"{c:4d}  {name:12s} {where}")

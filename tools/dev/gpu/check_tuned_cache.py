"""Developer check (GPU box): in the generator-tuned step, do the generator's weights change, does `_version` advance, and are the
cached GEMM weight images rebuilt from the NEW weights?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_train import Args
from hfa_gp_amd.trainer import Trainer
from hfa_gp_amd.synthetic import look_at_label
from hfa_gp_amd import ops

dev = torch.device("cuda:0"); torch.manual_seed(0)
tr = Trainer(Args(), dev, mode="3dmm", lpips="none"); tr.tune_generator()
g = torch.Generator().manual_seed(1); B = 2
real = (0.5 * torch.randn(B, 3, 256, 256, generator=g)).clamp(-1, 1).to(dev); params = torch.randn(B, 76, generator=g).to(dev)
lab = look_at_label(1.57 + 0.3 * torch.randn(B, generator=g), 1.57 + 0.15 * torch.randn(B, generator=g), flipped=False).to(dev)
gen = tr.gen.generator
w = gen.backbone.synthesis.b64.conv1.weight
calls = {"n": 0}
orig = ops.weight_prep_prec
def counting(*a, **k):
    calls["n"] += 1
    return orig(*a, **k)
ops.weight_prep_prec = counting
for step in range(3):
    w0, v0 = w.detach().clone(), w._version
    n0 = calls["n"]
    tr.gen_update(real, lab.clone(), params)
    torch.cuda.synchronize()
    img = gen._gemm_image(w)
    fresh = orig(w.detach().contiguous(), gen._precision_of(w) if gen._precision_of(w) != "f16x2" else "f16x3")
    print(f"step {step}: |dw|max {float((w.detach() - w0).abs().max()):.3e}  version {v0} -> {w._version}  weight_prep_prec calls in the step {calls['n'] - n0}  "
          f"cached image == image of the current weights: {bool(torch.equal(img, fresh))}")

"""Developer (GPU box): pointwise_bwd on the super-resolution-sized tensors against the number of pixel chunks per sample."""
import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfa_gp_amd import ops
dev = torch.device("cuda:0")
for B, H, C in ((2, 512, 128), (1, 512, 128), (2, 256, 256), (2, 256, 128), (2, 128, 256)):
    x = torch.randn(B, H, H, C, device=dev); d = torch.randn(B, H, H, C, device=dev); s = torch.randn(B, C, device=dev)
    prod = dict(dcoef=torch.rand(B, C, device=dev) + 0.5, bias=torch.randn(C, device=dev), noise=torch.randn(H, H, device=dev),
                noise_strength=0.1, act="lrelu", alpha=0.2, gain=math.sqrt(2.0), clamp=256.0)
    row = []
    for ch in (128, 256, 512, 1024, 2048, 4096):
        ops._DEV_PW_CHUNKS = str(ch)
        for _ in range(3):
            ops.pointwise_bwd(x, dxs_conv=d, s_conv=s, producer=prod)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.pointwise_bwd(x, dxs_conv=d, s_conv=s, producer=prod)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        row.append(f"{ch}: {us:6.1f} us ({3 * x.numel() * 4 / us / 1e6:5.2f} TB/s)")
    print(f"B={B} {H}^2 x {C}:  " + "  ".join(row))

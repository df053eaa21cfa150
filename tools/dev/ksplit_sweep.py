"""Developer (GPU box): split-K sweep of the small layers (conv + its reducer, back-to-back launches, HIP events).
usage: ksplit_sweep.py [B]"""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfa_gp_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
LAYERS = ((4, 512, 512), (8, 512, 512), (16, 512, 512), (32, 512, 512), (64, 512, 512), (64, 512, 256))
if len(sys.argv) > 2 and sys.argv[2] == "big":
    LAYERS = ((128, 512, 256), (128, 256, 256), (256, 256, 128), (256, 128, 128))
for H, cin, cout in LAYERS:
    for up in (1, 2):
        Hin = H // 2 if up == 2 else H
        if Hin < 4:
            continue
        x = torch.randn(B, Hin, Hin, cin, device=dev, generator=g)
        w = torch.randn(cout, cin, 3, 3, device=dev, generator=g) / 68
        wt = ops.weight_prep_prec(w, "f16x3")
        styles = torch.randn(B, cin, device=dev, generator=g)
        dcoef = torch.rand(B, cout, device=dev, generator=g)
        bias = torch.randn(cout, device=dev, generator=g)
        row = []
        for ks in (0, 1, 2, 4, 8, 16, 32):
            def run():
                if up == 2:
                    return ops.modconv(x, wt, cout, ops.CONVT3X3_UP2, styles=styles, ksplit=ks)
                return ops.modconv(x, wt, cout, ops.CONV3X3, styles=styles, dcoef=dcoef, bias=bias, act="lrelu", gain=math.sqrt(2), ksplit=ks)
            try:
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    run()
                e1.record()
                torch.cuda.synchronize()
                row.append(f"ks{ks}: {e0.elapsed_time(e1) / 50 * 1e3:6.1f}")
            except RuntimeError as e:
                row.append(f"ks{ks}: n/a")
        print(f"B={B} out {H}^2 {cin}->{cout} up={up}  " + "  ".join(row) + "  us")

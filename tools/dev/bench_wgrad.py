"""Developer micro-benchmark of the weight-gradient GEMM (GPU box). usage: B H Cin Cout [up] [bf16x3]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfa_gp_amd import ops
B, H, cin, cout = [int(v) for v in sys.argv[1:5]]
up = "up" in sys.argv[5:]
dev = torch.device("cuda:0")
x = torch.randn(B, H, H, cin, device=dev); s = torch.randn(B, cin, device=dev)
w = torch.randn(cout, cin, 3, 3, device=dev)
g = torch.randn(2, 2, B, H + 1, H + 1, cout, device=dev) if up else torch.randn(B, H, H, cout, device=dev)
mode = ops.CONVT3X3_UP2 if up else ops.CONV3X3
prec = "bf16x3" if "bf16x3" in sys.argv[5:] else "fp32"
for _ in range(2):
    ops.conv_wgrad(x, s, g, w, mode, precision=prec)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.conv_wgrad(x, s, g, w, mode, precision=prec)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
fl = 2.0 * B * H * H * cin * cout * 9
print(f"wgrad {prec} B={B} H={H} {cin}->{cout} up={up}: {ms*1e3:.0f} us, {fl/ms/1e9:.1f} TFLOP/s ({fl/ms/1e9/157.3:.3f})")

"""Developer micro-benchmark of one modulated-conv layer (run on the GPU box).
usage: bench_conv.py B H Cin Cout up [ksplit] [iters] [fp32|bf16x3|bf16x6|f16x3|f16]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfa_gp_amd import ops  # noqa: E402


def main():
    B, H, cin, cout, up = [int(v) for v in sys.argv[1:6]]
    ksplit = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    iters = int(sys.argv[7]) if len(sys.argv) > 7 else 10
    prec = sys.argv[8] if len(sys.argv) > 8 else "fp32"
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(B, H, H, cin, device=dev, generator=g) * float(os.environ.get("BENCH_XSCALE", "1"))
    w = torch.randn(cout, cin, 3, 3, device=dev, generator=g) * float(os.environ.get("BENCH_WSCALE", "1"))
    wt32, wsq = ops.weight_prep(w)
    wt = wt32 if prec == "fp32" else ops.weight_prep_prec(w, prec)
    styles = torch.randn(B, cin, device=dev, generator=g)
    dcoef = torch.rand(B, cout, device=dev, generator=g)
    bias = torch.randn(cout, device=dev, generator=g)
    mode = ops.CONVT3X3_UP2 if up == 2 else ops.CONV3X3

    def run(wt=wt):
        if up == 2:
            return ops.modconv(x, wt, cout, mode, styles=styles, ksplit=ksplit)
        return ops.modconv(x, wt, cout, mode, styles=styles, dcoef=dcoef, bias=bias, act="lrelu",
                           gain=math.sqrt(2), ksplit=ksplit)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        y = run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * B * H * H * cin * cout * 9
    err = ""
    if prec != "fp32":
        ref = run(wt32).double()
        err = f", rel. L2 error vs exact-fp32 kernel {float((y.double() - ref).norm() / ref.norm()):.2e}"
    print(f"conv {prec} B={B} H={H} {cin}->{cout} up={up} ksplit={ksplit}: {ms*1e3:.1f} us, {flops/ms/1e9:.1f} TFLOP/s "
          f"({flops/ms/1e9/157.3:.3f} of fp32 MFMA peak){err}")


if __name__ == "__main__":
    main()

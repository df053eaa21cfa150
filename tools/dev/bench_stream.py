"""Developer micro-benchmark of the streaming kernels (toRGB, upfir epilogue) on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfa_gp_amd import ops  # noqa: E402


def timeit(fn, iters=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    for h, cin in ((512, 128), (256, 256)):
        x = torch.randn(B, h, h, cin, device=dev)
        w = torch.randn(3, cin, device=dev)
        st = torch.randn(B, cin, device=dev)
        bias = torch.randn(3, device=dev)
        rgb = torch.randn(B, 3, h // 2, h // 2, device=dev)
        gb = x.numel() * 4 / 1e9
        for name, rin in (("with skip", rgb), ("no skip", None)):
            ms = timeit(lambda: ops.torgb_small(x, w, st, bias, rin, 256.0))
            print(f"torgb {h}^2 x {cin} B={B} {name}: {ms*1e3:.1f} us, {gb/ms*1e3:.0f} GB/s read")
        ms = timeit(lambda: x.sum())
        print(f"  torch sum of the same tensor: {ms*1e3:.1f} us, {gb/ms*1e3:.0f} GB/s")
        del x
    for h, c in ((256, 128), (128, 256)):
        yt = torch.randn(B, 2 * h + 1, 2 * h + 1, c, device=dev)
        dc = torch.rand(B, c, device=dev)
        bias = torch.randn(c, device=dev)
        gb = (yt.numel() + B * 4 * h * h * c) * 4 / 1e9
        ms = timeit(lambda: ops.upfir_epilogue(yt, dc, None, 0.0, bias, clamp=256.0))
        print(f"upfir {h}->{2*h} x {c} B={B}: {ms*1e3:.1f} us, {gb/ms*1e3:.0f} GB/s (in+out)")
        y = torch.empty(B, 2 * h, 2 * h, c, device=dev)
        ms = timeit(lambda: y.copy_(yt[:, :2 * h, :2 * h]))
        print(f"  torch strided copy of the same size: {ms*1e3:.1f} us, {gb/ms*1e3:.0f} GB/s")
        del yt, y


if __name__ == "__main__":
    main()

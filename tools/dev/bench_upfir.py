"""Developer bench (GPU box): the up-sampling layers of ffhq512_128 — fused one-pass kernel (csrc/upconv_fir.hip) against the
two-kernel form (transposed conv -> HBM -> FIR epilogue), per layer and for the whole synthesis."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfa_gp_amd import ops  # noqa: E402
from hfa_gp_amd.config import ffhq512_128  # noqa: E402
from hfa_gp_amd.generator import TriPlaneGenerator  # noqa: E402
from hfa_gp_amd.synthetic import make_inputs  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print(f"B={B} {prec}")
LAYERS = [(128, 32, 256), (256, 256, 128), (128, 256, 128), (64, 512, 256), (32, 512, 512), (16, 512, 512)]
if os.environ.get("UPFIR_LAYERS"):
    LAYERS = [tuple(int(v) for v in t.split(",")) for t in os.environ["UPFIR_LAYERS"].split(";")]
for (h, cin, cout) in LAYERS:
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, h, h, cin, generator=g).to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).to(dev)
    s = (torch.randn(B, cin, generator=g) + 1).to(dev)
    d = (torch.rand(B, cout, generator=g) + 0.5).to(dev)
    bias = torch.randn(cout, generator=g).to(dev)
    noise = torch.randn(2 * h, 2 * h, generator=g).to(dev)
    wt = ops.weight_prep_prec(w, prec)
    gf = 2.0 * B * h * h * cin * cout * 9 / 1e9

    def two():
        yt = ops.modconv(x, wt, cout, ops.CONVT3X3_UP2, styles=s)
        return ops.upfir_epilogue(yt, d, noise, 0.1, bias, clamp=256.0)

    def conv_only():
        return ops.modconv(x, wt, cout, ops.CONVT3X3_UP2, styles=s)

    t2, tc = timeit(two), timeit(conv_only)
    line = f"{cin:4d}->{cout:4d} @{h:3d}^2: two-kernel {t2:7.3f} ms (conv {tc:7.3f} + FIR {t2 - tc:6.3f}) = {gf / t2:6.1f} TF"
    if ops.upconv_fir_supported(x, wt, cout):
        for nseg in ([None] if len(sys.argv) <= 3 else [None] + [int(v) for v in sys.argv[3].split(",")]):
            if nseg is None:
                os.environ.pop("HFAGP_DEV_FIR_NSEG", None)
            else:
                os.environ["HFAGP_DEV_FIR_NSEG"] = str(nseg)
            tf = timeit(lambda: ops.upconv_fir(x, wt, cout, s, d, noise, 0.1, bias, clamp=256.0))
            line += f" | fused[nseg={nseg}] {tf:7.3f} ms = {gf / tf:6.1f} TF"
        os.environ.pop("HFAGP_DEV_FIR_NSEG", None)
        err = (ops.upconv_fir(x, wt, cout, s, d, noise, 0.1, bias, clamp=256.0) - two()).abs().max().item() if not os.environ.get("UPFIR_NOSYN") else -1
        line += f" | max diff {err:.1e}"
    else:
        line += " | fused: unsupported"
    print(line, flush=True)

if os.environ.get("UPFIR_NOSYN"):
    sys.exit(0)
cfg = ffhq512_128()
gen = TriPlaneGenerator(cfg, seed=0).requires_grad_(False).to(dev)
ws, c, us, ui = [t.to(dev) for t in make_inputs(cfg, B)]
for sr, store in ((None, "f32"), ("f16", "f32"), ("f16", "f16")):
    gen.sr_conv_precision, gen.sr_storage = sr, store
    for fuse in ("0", "auto", "1"):
        gen.fuse_up_fir = fuse
        t = timeit(lambda: gen.synthesis(ws, c, u_strat=us, u_imp=ui)["image"], n=8)
        print(f"synthesis B={B} sr_conv_precision={sr} sr_storage={store} fuse_up_fir={fuse}: {t:.2f} ms/step = {B / t * 1e3:.0f} frames/s",
              flush=True)

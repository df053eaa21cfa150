"""Developer timing of the ray-march backward (GPU box): sort + gather form (rows) against the scatter kernels.
usage: bench_raybwd.py [B] [iters] [rows|cols|both] [dec]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfa_gp_amd.synthetic import make_inputs
from hfa_gp_amd import ops
from hfa_gp_amd.config import ffhq512_128
from hfa_gp_amd.generator import TriPlaneGenerator

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
which = sys.argv[3] if len(sys.argv) > 3 else "both"
dec = len(sys.argv) > 4 and sys.argv[4] == "dec"
dev = torch.device("cuda:0")
cfg = ffhq512_128()
gen = TriPlaneGenerator(cfg, seed=0).to(dev)
ws, c, us, ui = [t.to(dev) for t in make_inputs(cfg, B)]
with torch.no_grad():
    planes = gen.backbone_planes(ws)
    if os.environ.get("PLANES_SCALE"):                 # developer: 0 -> constant density, no clustering of the importance samples
        planes = planes * float(os.environ["PLANES_SCALE"])
    u_s, u_i = gen._uniforms(B, dev, us, ui)
    g = torch.randn(B, 128 * 128, 32, device=dev)
    kw = gen._render_args(c)
    pam = getattr(gen, "_planes_absmax", None)
    res = {}
    for rows in {"rows": (True,), "cols": (False,), "both": (False, True, False, True)}[which]:
        call = lambda: ops.raymarch_bwd(g, planes, u_strat=u_s, u_imp=u_i, planes_absmax=pam, rows=rows, decoder_grads=dec, **kw)
        for _ in range(2):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            d = call()
        e1.record()
        torch.cuda.synchronize()
        res[rows] = d
        dp = d[0] if dec else d
        print(f"raymarch_bwd B={B} rows={rows} dec={dec}: {e0.elapsed_time(e1)/iters:.3f} ms/call ({e0.elapsed_time(e1)/iters/B:.3f} ms/frame), "
              f"checksum {dp.double().abs().sum().item():.4f}")
    if len(res) == 2:
        a, b = (res[True][0], res[False][0]) if dec else (res[True], res[False])
        print(f"max |rows - scatter| = {(a - b).abs().max().item():.3e} (scale {b.abs().max().item():.3e}), "
              f"rel L2 {((a - b).norm() / b.norm()).item():.3e}")
        if dec:
            for x, y, n in zip(res[True][1], res[False][1], ("w0", "b0", "w1", "b1")):
                print(f"  d {n}: max diff {(x - y).abs().max().item():.3e} (scale {y.abs().max().item():.3e})")

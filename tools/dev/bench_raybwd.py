"""Developer timing of the ray-march backward (GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfa_gp_amd.synthetic import make_inputs
from hfa_gp_amd import ops
from hfa_gp_amd.config import ffhq512_128
from hfa_gp_amd.generator import TriPlaneGenerator

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
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
    for two in ((False, True, False, True) if len(sys.argv) <= 3 else (sys.argv[3] == "2",)):
        for _ in range(2):
            ops.raymarch_bwd(g, planes, u_strat=u_s, u_imp=u_i, planes_absmax=pam, two_kernel=two, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            d = ops.raymarch_bwd(g, planes, u_strat=u_s, u_imp=u_i, planes_absmax=pam, two_kernel=two, **kw)
        e1.record()
        torch.cuda.synchronize()
        res[two] = d
        print(f"raymarch_bwd B={B} two_kernel={two}: {e0.elapsed_time(e1)/iters:.3f} ms/call ({e0.elapsed_time(e1)/iters/B:.3f} ms/frame), "
              f"checksum {d.double().abs().sum().item():.4f}")
    if len(res) == 2:
        diff = (res[True] - res[False]).abs().max().item()
        print(f"max |two-kernel - fused| = {diff:.3e} (scale {res[False].abs().max().item():.3e})")

"""Developer benchmark (GPU box): synthesis frames/s at B = 32 per conv precision.  usage: bench_prec.py [B] [iters] prec..."""
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfa_gp_amd.config import ffhq512_128  # noqa: E402
from hfa_gp_amd.generator import TriPlaneGenerator  # noqa: E402
from hfa_gp_amd.synthetic import make_inputs  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    precs = sys.argv[3:] or ["f16x3", "f16x2", "f16x3", "f16x2"]
    dev = torch.device("cuda:0")
    ws, c, us, ui = [t.to(dev) for t in make_inputs(ffhq512_128(), B)]
    base = None
    for prec in precs:
        cfg = dataclasses.replace(ffhq512_128(), conv_precision=prec)
        gen = TriPlaneGenerator(cfg, seed=0).to(dev)
        with torch.no_grad():
            for _ in range(3):
                out = gen.synthesis(ws, c, noise_mode="const", u_strat=us, u_imp=ui)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                out = gen.synthesis(ws, c, noise_mode="const", u_strat=us, u_imp=ui)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        img = out["image"].float()
        if base is None:
            base = img
        d = (img - base)
        print(f"{prec}: {ms:.2f} ms/step = {B / ms * 1e3:.1f} frames/s; vs first: mse {d.pow(2).mean().item():.2e} max {d.abs().max().item():.2e}")


if __name__ == "__main__":
    main()

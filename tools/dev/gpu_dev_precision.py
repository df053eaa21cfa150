"""Developer report: error of the 512^2 image vs the CPU oracle for each conv precision (run on the GPU box)."""
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfa_gp_amd.config import ffhq512_128  # noqa: E402
from hfa_gp_amd.generator import TriPlaneGenerator  # noqa: E402
from oracle import eg3d_oracle as O  # noqa: E402
from hfa_gp_amd.synthetic import make_inputs, perturb_state, state_cpu  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = ffhq512_128()
    gen = perturb_state(TriPlaneGenerator(cfg, seed=0))
    P = state_cpu(gen)
    gen = gen.to(dev).requires_grad_(False)
    ws, c, us, ui = make_inputs(cfg, 1)
    ref = O.synthesis(P, cfg, ws, c, us, ui)["image"]
    for prec, sr in (("fp32", None), ("bf16x6", None), ("f16x3", None), ("bf16x3", None), ("bf16x3", "f16"), ("f16", None)):
        gen.conv_precision, gen.sr_conv_precision = prec, sr
        prec = prec if sr is None else f"{prec}+{sr} SR"
        out = gen.synthesis(ws.to(dev), c.to(dev), u_strat=us.to(dev), u_imp=ui.to(dev))["image"].cpu()
        err = (out - ref).abs()
        print(f"{prec:14s}: max abs {err.max().item():.2e}, rms {err.pow(2).mean().sqrt().item():.2e}, "
              f"mse {err.pow(2).mean().item():.2e} (image range [{ref.min().item():.2f}, {ref.max().item():.2f}])")


if __name__ == "__main__":
    main()

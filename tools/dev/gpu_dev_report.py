"""Developer report (run on the GPU box): per-stage error of the HIP path vs the CPU oracle.
Not a test; prints max-abs / rms errors so tolerances in test_gpu_parity.py are set from measurement."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfa_gp_amd.synthetic import make_inputs, perturb_state, state_cpu  # noqa: E402
from hfa_gp_amd import ops  # noqa: E402
from hfa_gp_amd.config import PRESETS  # noqa: E402
from hfa_gp_amd.generator import TriPlaneGenerator  # noqa: E402
from oracle import eg3d_oracle as O  # noqa: E402


def err(a, b):
    d = (a.float().cpu() - b.float().cpu())
    return f"max {d.abs().max().item():.3e} rms {d.pow(2).mean().sqrt().item():.3e} (ref rms {b.float().pow(2).mean().sqrt().item():.3e})"


def main():
    names = sys.argv[1:] or ["tiny64", "small128"]
    dev = torch.device("cuda:0")
    for name in names:
        cfg = PRESETS[name]()
        B = 2 if name != "ffhq512_128" else 1
        gen = perturb_state(TriPlaneGenerator(cfg, seed=0))
        P = state_cpu(gen)
        gen = gen.to(dev)
        ws, c, us, ui = make_inputs(cfg, B)
        t = time.time()
        ref = O.synthesis(P, cfg, ws, c, us, ui, return_planes=True)
        t_cpu = time.time() - t
        out = gen.synthesis(ws.to(dev), c.to(dev), u_strat=us.to(dev), u_imp=ui.to(dev), return_planes=True)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(3):
            out = gen.synthesis(ws.to(dev), c.to(dev), u_strat=us.to(dev), u_imp=ui.to(dev), return_planes=True)
        torch.cuda.synchronize()
        t_gpu = (time.time() - t) / 3
        print(f"== {name} B={B}: oracle {t_cpu:.2f}s, hip {t_gpu*1e3:.2f} ms/call (eager)")
        planes_ref = ref["planes"].reshape(B, 3, 32, *ref["planes"].shape[-2:]).permute(0, 1, 3, 4, 2)
        print("  planes      ", err(out["planes"], planes_ref))
        print("  feature_img ", err(out["feature_image"].permute(0, 3, 1, 2), ref["feature_image"]))
        print("  image_raw   ", err(out["image_raw"], ref["image_raw"]))
        print("  image_depth ", err(out["image_depth"], ref["image_depth"]))
        print("  image       ", err(out["image"], ref["image"]))
        # renderer alone on the ORACLE's planes (isolates the ray-march kernel)
        feat, depth, wsum, tmm = gen.render(planes_ref.contiguous().to(dev), c.to(dev), us.to(dev), ui.to(dev))
        res = cfg.neural_rendering_resolution
        print("  raymarch|oracle planes", err(feat.view(B, res, res, 32).permute(0, 3, 1, 2), ref["feature_image"]))
        # SR alone on the oracle's feature image
        fi = ref["feature_image"].permute(0, 2, 3, 1).contiguous().to(dev)
        img = gen.superres(ref["image_raw"].contiguous().to(dev), fi, ws.to(dev))
        print("  superres|oracle feat  ", err(img, ref["image"]))
        mse = (out["image"].cpu() - ref["image"]).pow(2).mean().item()
        print(f"  image MSE {mse:.3e}")


if __name__ == "__main__":
    with torch.no_grad():
        main()

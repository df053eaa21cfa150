"""Developer micro-benchmark of the ray-march kernel alone (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfa_gp_amd.synthetic import make_inputs  # noqa: E402
from hfa_gp_amd.config import ffhq512_128  # noqa: E402
from hfa_gp_amd.generator import TriPlaneGenerator  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda:0")
    cfg = ffhq512_128()
    gen = TriPlaneGenerator(cfg, seed=0).to(dev)
    ws, c, us, ui = [t.to(dev) for t in make_inputs(cfg, B)]
    with torch.no_grad():
        planes = gen.backbone_planes(ws)
        for _ in range(2):
            gen.render(planes, c, us, ui)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            out = gen.render(planes, c, us, ui)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    gb = B * 2.424438784   # GB per launch
    print(f"raymarch B={B}: {ms:.3f} ms/launch = {ms / B * 1e3:.1f} us/frame, {gb / ms * 1e3:.0f} GB/s algorithmic "
          f"({gb / ms * 1e3 / 8000:.3f} of 8 TB/s); checksum {out[0].double().sum().item():.6f}")


if __name__ == "__main__":
    main()

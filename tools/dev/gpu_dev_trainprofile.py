"""Developer report: host-side (Python) profile of the fitting step (run on the GPU box)."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_train import Args  # noqa: E402
from hfa_gp_amd.trainer import Trainer  # noqa: E402
from hfa_gp_amd.synthetic import look_at_label  # noqa: E402


def main():
    B = 2
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tr = Trainer(Args(), dev, mode="3dmm")
    g = torch.Generator().manual_seed(1)
    real = (0.5 * torch.randn(B, 3, 256, 256, generator=g)).clamp(-1, 1).to(dev)
    params = torch.randn(B, 76, generator=g).to(dev)
    label0 = look_at_label(1.57 + 0.3 * torch.randn(B, generator=g), 1.57 + 0.15 * torch.randn(B, generator=g), flipped=False).to(dev)
    for _ in range(3):
        tr.gen_update(real, label0.clone(), params)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        tr.gen_update(real, label0.clone(), params)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(22)


if __name__ == "__main__":
    main()

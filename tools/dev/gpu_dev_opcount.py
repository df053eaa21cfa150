"""Developer report (GPU box): which torch ops launch the small kernels of one fitting step?  usage: gpu_dev_opcount.py [rgb|3dmm]"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_train import Args  # noqa: E402
from hfa_gp_amd.trainer import Trainer  # noqa: E402
from hfa_gp_amd.synthetic import look_at_label  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "rgb"
    B = 2
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tr = Trainer(Args(), dev, mode=mode, lpips="none")
    if len(sys.argv) > 2 and sys.argv[2] == "tuned":
        tr.tune_generator()
    g = torch.Generator().manual_seed(1)
    real = (0.5 * torch.randn(B, 3, 256, 256, generator=g)).clamp(-1, 1).to(dev)
    params = torch.randn(B, 76, generator=g).to(dev)
    label0 = look_at_label(1.57 + 0.3 * torch.randn(B, generator=g), 1.57 + 0.15 * torch.randn(B, generator=g), flipped=False).to(dev)
    step = (lambda: tr.gen_update(real, label0.clone())) if mode == "rgb" else (lambda: tr.gen_update(real, label0.clone(), params))
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    n = 5
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
        for _ in range(n):
            step()
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages():
        dev_us = getattr(e, "device_time_total", None)
        if dev_us is None:
            dev_us = getattr(e, "cuda_time_total", 0.0)
        if e.key.startswith("aten::") and dev_us > 0:
            rows.append((e.count / n, dev_us / n, e.key))
    rows.sort(key=lambda r: -r[0])
    kern = {}
    for e in prof.events():
        if getattr(e, "device_type", None) is not None and str(e.device_type).endswith("CUDA") and not e.name.startswith(("hfagp", "void hfagp", "Cijk")):
            k = kern.setdefault(e.name[:90], [0, 0.0]); k[0] += 1; k[1] += e.device_time if hasattr(e, "device_time") else 0.0
    for name, (c, us) in sorted(kern.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"{c / n:10.1f} {us / n:15.1f}  kernel {name}")
    print(f"{'calls/step':>10s} {'device us/step':>15s}  op")
    for c, us, k in rows[:45]:
        print(f"{c:10.1f} {us:15.1f}  {k}")


if __name__ == "__main__":
    main()

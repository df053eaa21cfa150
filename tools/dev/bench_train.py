"""Developer timing of one fitting step (BASELINE config 3 mechanics) on the GPU box."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfa_gp_amd import headnerf
from hfa_gp_amd.trainer import Trainer
from hfa_gp_amd.synthetic import look_at_label


class Args:
    out_pose = False; person_2 = False; params_len = 76; size = 256; batch_size = 1; lr = 3e-4
    latent_dim_style = 512; latent_dim_shape = 50; generator_preset = "ffhq512_128"; generator_seed = 0


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    mode = sys.argv[3] if len(sys.argv) > 3 else "3dmm"
    tr = Trainer(Args(), dev, mode=mode, lpips="none")
    if len(sys.argv) > 4 and sys.argv[4] == "tuned":      # the reference's regime after tune_iter (trainer_rgb.py:69-71)
        tr.tune_generator()
    g = torch.Generator().manual_seed(1)
    real = (0.5 * torch.randn(B, 3, 256, 256, generator=g)).clamp(-1, 1).to(dev)
    params = torch.randn(B, 76, generator=g).to(dev)
    label0 = look_at_label(1.57 + 0.3 * torch.randn(B, generator=g), 1.57 + 0.15 * torch.randn(B, generator=g), flipped=False).to(dev)
    step = (lambda: tr.gen_update(real, label0.clone())) if mode == "rgb" else (lambda: tr.gen_update(real, label0.clone(), params))
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)]
    t = time.perf_counter()
    evs[0].record()
    for i in range(iters):
        l2 = step()[-3]
        evs[i + 1].record()
    t_enq = (time.perf_counter() - t) / iters           # host time to ENQUEUE a step (nothing synchronised yet)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / iters
    per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(iters))
    print(f"train step B={B}: {dt*1e3:.2f} ms/step ({dt/B*1e3:.2f} ms/frame), per-step HIP events median {per[len(per)//2]:.2f} "
          f"min {per[0]:.2f} max {per[-1]:.2f} ms, host enqueue {t_enq*1e3:.2f} ms/step, l2={float(l2):.4f}, mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")


if __name__ == "__main__":
    main()

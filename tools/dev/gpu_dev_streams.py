"""Developer experiment: render B frames as one call vs two half-batches on two HIP streams (memory-bound epilogue /
FIR kernels of one half under the MFMA-bound convs of the other)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfa_gp_amd.synthetic import make_inputs
from hfa_gp_amd.config import ffhq512_128
from hfa_gp_amd.generator import TriPlaneGenerator

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nstreams = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
cfg = ffhq512_128()
gen = TriPlaneGenerator(cfg, seed=0).requires_grad_(False).to(dev)
ws, c, us, ui = [t.to(dev) for t in make_inputs(cfg, B)]
r = cfg.neural_rendering_resolution ** 2
streams = [torch.cuda.Stream() for _ in range(nstreams)]
h = B // nstreams
parts = [(ws[i * h:(i + 1) * h].contiguous(), c[i * h:(i + 1) * h].contiguous(), us[i * h:(i + 1) * h].contiguous(),
          ui[i * h * r:(i + 1) * h * r].contiguous()) for i in range(nstreams)]


def one():
    gen.synthesis(ws, c, noise_mode="const", u_strat=us, u_imp=ui)


def split():
    for s, (w_, c_, us_, ui_) in zip(streams, parts):
        with torch.cuda.stream(s):
            gen.synthesis(w_, c_, noise_mode="const", u_strat=us_, u_imp=ui_)


for name, fn in (("one call", one), (f"{nstreams} streams", split), ("one call", one), (f"{nstreams} streams", split)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print(f"B={B} {name}: {dt*1e3:.2f} ms/step, {B/dt:.1f} frames/s")

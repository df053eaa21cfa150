"""Developer: is one synthesis call host-bound?  Enqueue time (host returns from the call, nothing synchronised) against the
device time of the same calls, batch B, with and without the image side stream."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfa_gp_amd.synthetic import make_inputs, perturb_state
from hfa_gp_amd.config import ffhq512_128
from hfa_gp_amd.generator import TriPlaneGenerator

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
cfg = ffhq512_128()
gen = perturb_state(TriPlaneGenerator(cfg, seed=0)).requires_grad_(False).to(dev)
ws, c, us, ui = [t.to(dev) for t in make_inputs(cfg, B)]
for side in (0, 4, 0, 4):
    gen.side_stream_max_batch = side
    for _ in range(5):
        gen.synthesis(ws, c, noise_mode="const", u_strat=us, u_imp=ui)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        gen.synthesis(ws, c, noise_mode="const", u_strat=us, u_imp=ui)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B} side_stream_max_batch={side}: enqueue {1e3 * (t1 - t0) / n:.3f} ms/call, wall {1e3 * (t2 - t0) / n:.3f} ms/call")

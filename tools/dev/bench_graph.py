"""Developer check: hipGraph capture of the whole synthesis (GPU box)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfa_gp_amd.synthetic import make_inputs
from hfa_gp_amd.config import ffhq512_128
from hfa_gp_amd.generator import TriPlaneGenerator

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
cfg = ffhq512_128()
gen = TriPlaneGenerator(cfg, seed=0).requires_grad_(False).to(dev)
ws, c, us, ui = [t.to(dev) for t in make_inputs(cfg, B)]

def step():
    return gen.synthesis(ws, c, u_strat=us, u_imp=ui)["image"]

for _ in range(3):
    ref = step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
eager = (time.perf_counter() - t) / 20
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
g.replay()
torch.cuda.synchronize()
print("graph vs eager max diff", (out - ref).abs().max().item())
t = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t) / 20
print(f"B={B}: eager {eager*1e3:.2f} ms/step ({B/eager:.0f} fps), graph {graph*1e3:.2f} ms/step ({B/graph:.0f} fps)")

import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from hfa_gp_amd.trainer import MultiTensorAdam
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(12)
for s in [(), (3,), (5, 7)]:
    a = torch.nn.Parameter(torch.randn(s, generator=g).to(dev)); b = torch.nn.Parameter(a.detach().clone())
    o1 = MultiTensorAdam([a], lr=3e-4); o2 = torch.optim.Adam([b], lr=3e-4, foreach=False, fused=False)
    for step in range(3):
        gr = (torch.randn(tuple(a.shape), generator=g) * (10.0 ** (step - 1))).to(dev)
        a.grad, b.grad = gr.clone(), gr.clone()
        o1.step(); o2.step()
        print(s, step, "p", a.detach().flatten()[:2].tolist(), b.detach().flatten()[:2].tolist(), "m", o1.state[a]["exp_avg"].flatten()[:2].tolist(), o2.state[b]["exp_avg"].flatten()[:2].tolist(),
              "v", o1.state[a]["exp_avg_sq"].flatten()[:1].tolist(), o2.state[b]["exp_avg_sq"].flatten()[:1].tolist(), "step", float(o1.state[a]["step"]), float(o2.state[b]["step"]))

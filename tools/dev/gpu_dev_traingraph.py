"""Developer experiment: the fitting step (fwd + bwd + Adam) captured in a HIP graph (run on the GPU box).
usage: gpu_dev_traingraph.py [B] [iters]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_train import Args
from hfa_gp_amd.trainer import Trainer
from hfa_gp_amd.synthetic import look_at_label


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tr = Trainer(Args(), dev, mode="3dmm")
    tr.g_optim = torch.optim.Adam([p for p in tr.gen.parameters() if p.requires_grad], lr=3e-4, capturable=True)
    g = torch.Generator().manual_seed(1)
    real = (0.5 * torch.randn(B, 3, 256, 256, generator=g)).clamp(-1, 1).to(dev)
    params = torch.randn(B, 76, generator=g).to(dev)
    label0 = look_at_label(1.57 + 0.3 * torch.randn(B, generator=g), 1.57 + 0.15 * torch.randn(B, generator=g), flipped=False).to(dev)
    label = label0.clone()

    def step():
        label.copy_(label0)
        tr.g_optim.zero_grad(set_to_none=False)
        tr.gen.train()
        generated = tr.gen(params, label, False)
        from hfa_gp_amd.trainer import pooled_l2
        l2, pooled = pooled_l2(tr.face_pool, real, generated, False)
        l2.backward()
        tr.g_optim.step()
        return l2

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            l2 = step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        l2 = step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t) / iters
    print(f"eager  B={B}: {eager*1e3:.2f} ms/step, l2 {float(l2):.5f}")
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        l2g = step()
    torch.cuda.synchronize()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        graph.replay()
    torch.cuda.synchronize()
    gr = (time.perf_counter() - t) / iters
    print(f"graph  B={B}: {gr*1e3:.2f} ms/step, l2 {float(l2g):.5f}")


if __name__ == "__main__":
    main()

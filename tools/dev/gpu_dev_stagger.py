"""Developer experiment (GPU box): two half-batches on two HIP streams, STAGGERED by a phase — the super-resolution convs
(matrix pipe) of one half run while the other half is in its backbone / ray march (vector ALU + L2 gather) — against one call
of the whole batch.  (tools/dev/gpu_dev_streams.py launched both halves in phase: GEMM met GEMM, ray march met ray march.)
usage: gpu_dev_stagger.py [B] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfa_gp_amd.config import ffhq512_128  # noqa: E402
from hfa_gp_amd.generator import TriPlaneGenerator  # noqa: E402
from hfa_gp_amd.synthetic import make_inputs  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda:0")
cfg = ffhq512_128()
gens = [TriPlaneGenerator(cfg, seed=0).requires_grad_(False).to(dev) for _ in range(2)]
ws, c, us, ui = [t.to(dev) for t in make_inputs(cfg, B)]
r = cfg.neural_rendering_resolution ** 2
h = B // 2
parts = [(ws[i * h:(i + 1) * h].contiguous(), c[i * h:(i + 1) * h].contiguous(),
          us[i * h:(i + 1) * h].reshape(h, r, -1).contiguous(), ui[i * h * r:(i + 1) * h * r].contiguous()) for i in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def front(g, p):          # backbone + ray march
    w_, c_, us_, ui_ = p
    planes = g.backbone_planes(w_)
    feat, depth, wsum, tmm = g.render(planes, c_, us_, ui_, planes_absmax=getattr(g, "_planes_absmax", None))
    feat_img = feat.view(h, cfg.neural_rendering_resolution, cfg.neural_rendering_resolution, 32)
    rgb_raw = feat_img[..., :3].permute(0, 3, 1, 2).contiguous()
    return rgb_raw, feat_img


def back(g, p, mid):      # super-resolution
    return g.superres(mid[0], mid[1], p[0])


def one_call(n):
    for _ in range(n):
        gens[0].synthesis(ws, c, noise_mode="const", u_strat=us, u_imp=ui)


def in_phase(n):
    for _ in range(n):
        for g, s, p in zip(gens, streams, parts):
            with torch.cuda.stream(s):
                back(g, p, front(g, p))


def staggered(n):
    """stream 0: F0 S0 F0 S0 ...; stream 1 starts its front part when stream 0 enters its first super-resolution, so that
    in steady state S(one half) always runs beside F(other half)."""
    with torch.no_grad():
        ev = None
        for k in range(n):
            with torch.cuda.stream(streams[0]):
                if ev is not None:
                    streams[0].wait_event(ev)          # do not run ahead: F0(k) starts when S1(k-1) starts
                m0 = front(gens[0], parts[0])
                e0 = torch.cuda.Event(); e0.record(streams[0])
                back(gens[0], parts[0], m0)
            with torch.cuda.stream(streams[1]):
                streams[1].wait_event(e0)              # F1(k) starts when S0(k) starts
                m1 = front(gens[1], parts[1])
                ev = torch.cuda.Event(); ev.record(streams[1])
                back(gens[1], parts[1], m1)


with torch.no_grad():
    for name, fn in (("one call", one_call), ("2 streams in phase", in_phase), ("2 streams staggered", staggered),
                     ("one call", one_call), ("2 streams staggered", staggered)):
        fn(3)
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn(steps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / steps
        print(f"B={B} {name}: {dt*1e3:.2f} ms/step, {B/dt:.1f} frames/s", flush=True)

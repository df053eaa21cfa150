"""Synthetic frame sets with a KNOWN optimum for the fitting configurations of BASELINE.json / SURVEY.md §8d
(config 3: 500 frames, RGB-driven; config 4: 2000 frames, 3DMM-driven, frame-sharded; config 5: audio features).

The reference reads its frames from `HeadData*` data sets (`/root/reference/code/dataset.py`: image [3,256,256] in
[-1,1], 25-float label with columns [1,2,5,6,9,10] pre-negated, optional 76-d 3DMM `params`); the data sets and their
preprocessing are out of scope, so frames are RENDERED here: a hidden "true" latent basis (seed 40), N random
coordinates alpha* and gaussian cameras (`trainer_rgb.py:28-29`: h, v ~ N(pi/2, 0.3 / 0.155), r = 2.7), targets =
`AdaptiveAvgPool2d(size)` of the generator's own render of `alpha* @ Q*^T + delta*`.  A fit therefore has loss 0 as its
optimum, and for the 3DMM variant the driver input is an exact linear code of alpha* (`params = alpha* @ M^+`), which
the activation-less `Weights_3DMM` stack can represent.

Everything stays on the device; nothing here is timed as part of the hot path.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .cam_utils import create_cam2world_matrix, make_label, sample_camera_positions


def gaussian_labels(n: int, device, seed: int = 20, r: float = 2.7) -> torch.Tensor:
    """n camera labels [n, 25] as the data set yields them (un-flipped), h ~ N(pi/2, 0.3), v ~ N(pi/2, 0.155)."""
    g = torch.Generator().manual_seed(seed)
    h = math.pi / 2 + 0.3 * torch.randn(n, generator=g)
    v = (math.pi / 2 + 0.155 * torch.randn(n, generator=g)).clamp(1e-5, math.pi - 1e-5)
    pos = torch.stack([r * torch.sin(v) * torch.cos(h), r * torch.cos(v), r * torch.sin(v) * torch.sin(h)], -1).to(device)
    return make_label(create_cam2world_matrix(-pos / r, pos, device=device))


@torch.no_grad()
def make_frame_set(model, n: int, size: int = 256, seed: int = 40, params_len: Optional[int] = None,
                   render_batch: int = 8, alpha_scale: float = 1.0, only: Optional[tuple] = None) -> Dict[str, torch.Tensor]:
    """`model`: a HeadNeRF_* module (its generator renders the targets; its own basis is NOT the hidden one).
    Returns {'real' [n,3,size,size], 'label' [n,25] un-flipped, 'alpha' [n,K], 'params' [n,P] | absent,
    'true_bases' [K,14*dim], 'true_delta' [14*dim]}.  `only = (lo, hi)`: render only the targets of frames [lo, hi) (a rank's
    shard; the other rows of 'real' stay zero — the frame set itself is identical on every rank)."""
    dev = next(model.parameters()).device
    k, dim = model.dim_shape, model.dim
    g = torch.Generator().manual_seed(seed)
    true_bases = torch.randn(k, 14 * dim, generator=g).to(dev)
    true_delta = true_bases.mean(0)
    alpha = (alpha_scale * torch.randn(n, k, generator=g)).to(dev)
    label = gaussian_labels(n, dev, seed=seed + 1)
    q = torch.linalg.qr((true_bases + 1e-8).T, mode="reduced")[0]
    ws = (alpha @ q.T).view(n, 14, dim) + true_delta.view(14, dim)
    lo, hi = only if only is not None else (0, n)
    real = torch.zeros(n, 3, size, size, device=dev)
    for i in range(lo, hi, render_batch):
        e = min(hi, i + render_batch)
        lab = label[i:e].clone()
        lab[:, [1, 2, 5, 6, 9, 10]] *= -1                     # what get_image feeds the generator (headnerf.py:132)
        img = model.generator.synthesis(ws[i:e].contiguous(), c=lab, noise_mode="const")["image"]
        real[i:e] = F.adaptive_avg_pool2d(img, (size, size)).clamp(-1, 1)
    out = {"real": real, "label": label, "alpha": alpha, "true_bases": true_bases, "true_delta": true_delta}
    if params_len is not None:
        m = torch.randn(params_len, k, generator=g).to(dev) / math.sqrt(params_len)      # alpha* = params @ M
        out["params"] = alpha @ torch.linalg.pinv(m)
        out["params_map"] = m
    return out


def audio_features(n: int, seed: int = 50) -> torch.Tensor:
    """DeepSpeech-shaped windows aud[N,16,29] ~ N(0,1)."""
    return torch.randn(n, 16, 29, generator=torch.Generator().manual_seed(seed))


# ----------------------------------------------------------------------------- seeded inputs of BASELINE config 2 (bench.py, smoke(), tests)
INTRINSICS = [4.2647, 0, 0.5, 0, 4.2647, 0.5, 0, 0, 1]   # /root/reference/code/trainer_rgb.py:32
FLIP_COLUMNS = [1, 2, 5, 6, 9, 10]                        # /root/reference/code/networks/headnerf.py:132


def look_at_label(h: torch.Tensor, v: torch.Tensor, r: float = 2.7, flipped: bool = True) -> torch.Tensor:
    """25-float labels for cameras on a sphere of radius r looking at the origin (same geometry as
    cam_utils.sample_camera_positions + create_cam2world_matrix, written out independently of them).  cam_utils labels have
    the camera z axis pointing AWAY from the origin; HeadNeRF.get_image negates columns [1,2,5,6,9,10] before the
    generator sees them (headnerf.py:132).  flipped=True returns what the GENERATOR is fed."""
    n = h.shape[0]
    pos = torch.stack([r * torch.sin(v) * torch.cos(h), r * torch.cos(v), r * torch.sin(v) * torch.sin(h)], -1)
    fwd = F.normalize(-pos, dim=-1)
    up = torch.tensor([0.0, 1.0, 0.0]).expand_as(fwd)
    left = F.normalize(torch.cross(up, fwd, dim=-1), dim=-1)
    up = F.normalize(torch.cross(fwd, left, dim=-1), dim=-1)
    m = torch.eye(4).repeat(n, 1, 1)
    m[:, :3, :3] = torch.stack((-left, up, -fwd), dim=-1)
    m[:, :3, 3] = pos
    label = torch.cat([m.reshape(n, 16), torch.tensor(INTRINSICS).repeat(n, 1)], -1)
    if flipped:
        label[:, FLIP_COLUMNS] *= -1
    return label


def make_inputs(cfg, batch: int, seed: int = 10):
    """BASELINE.md §3 inputs: ws ~ N(0,1), gaussian cameras around (pi/2, pi/2) (trainer_rgb.py:28-29), uniforms for the
    renderer — (ws [B,14,512], c [B,25] as the generator is fed, u_strat [B,R,Sc,1], u_imp [B*R,Sf]), CPU tensors."""
    g = torch.Generator().manual_seed(seed)
    ws = torch.randn(batch, cfg.num_ws, cfg.w_dim, generator=g)
    h = math.pi / 2 + 0.3 * torch.randn(batch, generator=g)
    v = math.pi / 2 + 0.155 * torch.randn(batch, generator=g)
    c = look_at_label(h, v)
    r = cfg.neural_rendering_resolution ** 2
    u_strat = torch.rand(batch, r, cfg.depth_resolution, 1, generator=g)
    u_imp = torch.rand(batch * r, cfg.depth_resolution_importance, generator=g)
    return ws, c, u_strat, u_imp


def perturb_state(gen, seed: int = 3):
    """Random-init leaves biases and noise_strength at 0; make them non-trivial so the bias / noise paths are exercised."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in gen.named_parameters():
            if name.endswith("noise_strength"):
                p.copy_(0.1 * torch.randn([], generator=g))
            elif name.endswith(".bias") and ".affine." not in name and "mapping" not in name:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    return gen


def state_cpu(gen):
    """The generator's state_dict as detached fp32 CPU tensors (EG3D key names)."""
    return {k: v.detach().float().cpu() for k, v in gen.state_dict().items()}

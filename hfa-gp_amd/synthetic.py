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

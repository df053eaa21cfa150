"""Shared helpers for the parity tests (seeded inputs, oracle access)."""
from __future__ import annotations

import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

INTRINSICS = [4.2647, 0, 0.5, 0, 4.2647, 0.5, 0, 0, 1]   # /root/reference/code/trainer_rgb.py:32


FLIP_COLUMNS = [1, 2, 5, 6, 9, 10]                        # /root/reference/code/networks/headnerf.py:132


def look_at_label(h: torch.Tensor, v: torch.Tensor, r: float = 2.7, flipped: bool = True) -> torch.Tensor:
    """25-float labels for cameras on a sphere of radius r looking at the origin (same geometry as
    cam_utils.sample_camera_positions + create_cam2world_matrix).  cam_utils labels have the camera
    z axis pointing AWAY from the origin; HeadNeRF.get_image negates columns [1,2,5,6,9,10] before the
    generator sees them (headnerf.py:132).  flipped=True returns what the GENERATOR is fed."""
    n = h.shape[0]
    pos = torch.stack([r * torch.sin(v) * torch.cos(h), r * torch.cos(v), r * torch.sin(v) * torch.sin(h)], -1)
    fwd = torch.nn.functional.normalize(-pos, dim=-1)
    up = torch.tensor([0.0, 1.0, 0.0]).expand_as(fwd)
    left = torch.nn.functional.normalize(torch.cross(up, fwd, dim=-1), dim=-1)
    up = torch.nn.functional.normalize(torch.cross(fwd, left, dim=-1), dim=-1)
    m = torch.eye(4).repeat(n, 1, 1)
    m[:, :3, :3] = torch.stack((-left, up, -fwd), dim=-1)
    m[:, :3, 3] = pos
    label = torch.cat([m.reshape(n, 16), torch.tensor(INTRINSICS).repeat(n, 1)], -1)
    if flipped:
        label[:, FLIP_COLUMNS] *= -1
    return label


def make_inputs(cfg, batch: int, seed: int = 10):
    """ws ~ N(0,1), gaussian cameras around (pi/2, pi/2) (trainer_rgb.py:28-29), uniforms for the renderer."""
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
    """Random-init leaves biases and noise_strength at 0; make them non-trivial so the
    bias / noise paths are exercised."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in gen.named_parameters():
            if name.endswith("noise_strength"):
                p.copy_(0.1 * torch.randn([], generator=g))
            elif name.endswith(".bias") and ".affine." not in name and "mapping" not in name:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
    return gen


def state_cpu(gen):
    return {k: v.detach().float().cpu() for k, v in gen.state_dict().items()}

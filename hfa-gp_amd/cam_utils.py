"""Camera label synthesis — counterpart of /root/reference/code/cam_utils.py:12-80 and of the
intrinsics literal in code/trainer_rgb.py:32.  A label is 16 cam2world floats (row-major 4x4)
followed by 9 normalised intrinsics.  Pinned by tests/golden (cam_* vectors)."""
from __future__ import annotations

import math
import random
from typing import Tuple

import torch

FFHQ_INTRINSICS = (4.2647, 0.0, 0.5, 0.0, 4.2647, 0.5, 0.0, 0.0, 1.0)   # trainer_rgb.py:32


def normalize_vecs(v: torch.Tensor) -> torch.Tensor:
    return v / torch.norm(v, dim=-1, keepdim=True)


def _angles(device, n, h_std, v_std, h_mean, v_mean, mode) -> Tuple[torch.Tensor, torch.Tensor]:
    def uni(scale):
        return (torch.rand((n, 1), device=device) - 0.5) * 2 * scale

    def gauss(scale):
        return torch.randn((n, 1), device=device) * scale

    if mode == "uniform":
        return uni(h_std) + h_mean, uni(v_std) + v_mean
    if mode in ("normal", "gaussian"):
        return gauss(h_std) + h_mean, gauss(v_std) + v_mean
    if mode == "hybrid":
        if random.random() < 0.5:
            return uni(h_std * 2) + h_mean, uni(v_std * 2) + v_mean
        return gauss(h_std) + h_mean, gauss(v_std) + v_mean
    if mode == "truncated_gaussian":
        # the reference calls an undefined `truncated_normal_` here (cam_utils.py:36-37)
        raise NotImplementedError("mode 'truncated_gaussian' is not runnable in the reference either")
    if mode == "spherical_uniform":
        theta = uni(h_std) + h_mean
        v = torch.clamp(uni(v_std / math.pi) + v_mean / math.pi, 1e-5, 1 - 1e-5)
        return theta, torch.arccos(1 - 2 * v)
    ones = torch.ones((n, 1), device=device, dtype=torch.float)
    return ones * h_mean, ones * v_mean


def sample_camera_positions(device, n=1, r=1, horizontal_stddev=1, vertical_stddev=1,
                            horizontal_mean=math.pi * 0.5, vertical_mean=math.pi * 0.5, mode="normal"):
    """n points on the sphere of radius r.  theta = yaw, phi = pitch (clamped to (0, pi)).
    Returns (points [n,3], phi, theta) like the reference."""
    theta, phi = _angles(device, n, horizontal_stddev, vertical_stddev, horizontal_mean, vertical_mean, mode)
    phi = torch.clamp(phi, 1e-5, math.pi - 1e-5)
    pts = torch.zeros((n, 3), device=device)
    pts[:, 0:1] = r * torch.sin(phi) * torch.cos(theta)
    pts[:, 2:3] = r * torch.sin(phi) * torch.sin(theta)
    pts[:, 1:2] = r * torch.cos(phi)
    return pts, phi, theta


def create_cam2world_matrix(forward_vector: torch.Tensor, origin: torch.Tensor, device=None) -> torch.Tensor:
    """Look-at matrix from a viewing direction and a camera position: columns (-left, up, -forward)."""
    fwd = normalize_vecs(forward_vector)
    up0 = torch.tensor([0, 1, 0], dtype=torch.float, device=device).expand_as(fwd)
    left = normalize_vecs(torch.cross(up0, fwd, dim=-1))
    up = normalize_vecs(torch.cross(fwd, left, dim=-1))
    n = fwd.shape[0]
    rot = torch.eye(4, device=device).unsqueeze(0).repeat(n, 1, 1)
    rot[:, :3, :3] = torch.stack((-left, up, -fwd), dim=-1)
    trans = torch.eye(4, device=device).unsqueeze(0).repeat(n, 1, 1)
    trans[:, :3, 3] = origin
    return trans @ rot


def make_label(cam2world: torch.Tensor, intrinsics=FFHQ_INTRINSICS) -> torch.Tensor:
    n = cam2world.shape[0]
    k = torch.tensor(intrinsics, dtype=cam2world.dtype, device=cam2world.device).reshape(1, -1).repeat(n, 1)
    return torch.cat((cam2world.reshape(n, -1), k), -1)


def cam_sampler(batch: int, device, horizontal_mean=0.5, vertical_mean=0.5, horizontal_stddev=0.3,
                vertical_stddev=0.155, r=2.7) -> torch.Tensor:
    """trainer_rgb.py:27-42 (`cam_sampler`, `cam_sampler_pose`): gaussian poses around the mean, label [B,25]."""
    pts, _, _ = sample_camera_positions(device, n=batch, r=r, horizontal_mean=horizontal_mean * math.pi,
                                        vertical_mean=vertical_mean * math.pi, horizontal_stddev=horizontal_stddev,
                                        vertical_stddev=vertical_stddev, mode="gaussian")
    return make_label(create_cam2world_matrix(-pts, pts, device=device))

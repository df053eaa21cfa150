"""LPIPS(net='alex') — the perceptual term of the reference's objective
(/root/reference/code/trainer_rgb.py:10,62,86-87: ``LPIPS(net='alex')`` on the 256^2 images).

The ``lpips`` package and its weights are not available offline (SURVEY.md §7 "hard parts"), so this is a
self-contained module with the SAME architecture and ``state_dict`` key names as ``lpips.LPIPS(net='alex')``
(``net.slice{1..5}.{0,3,6,8,10}.{weight,bias}``, ``lin{0..4}.model.1.weight``, ``scaling_layer.{shift,scale}``), so a
user-supplied ``lpips`` state dict loads with ``load_state_dict``.  Without weights it is random-initialised: the
step then has the right structure and cost, not the reference's numbers.  It runs on PyTorch-ROCm (MIOpen): it is
a loss on 256^2 images, not part of the generator hot path.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn


class _ScalingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scale", torch.tensor([.458, .448, .450])[None, :, None, None])

    def forward(self, x):
        return (x - self.shift) / self.scale


class _AlexFeatures(nn.Module):
    """torchvision AlexNet `features`, split after each ReLU like lpips.pretrained_networks.alexnet."""

    def __init__(self):
        super().__init__()
        self.slice1 = nn.Sequential()
        self.slice2 = nn.Sequential()
        self.slice3 = nn.Sequential()
        self.slice4 = nn.Sequential()
        self.slice5 = nn.Sequential()
        self.slice1.add_module("0", nn.Conv2d(3, 64, 11, 4, 2))
        self.slice1.add_module("1", nn.ReLU(inplace=False))
        self.slice2.add_module("2", nn.MaxPool2d(3, 2))
        self.slice2.add_module("3", nn.Conv2d(64, 192, 5, padding=2))
        self.slice2.add_module("4", nn.ReLU(inplace=False))
        self.slice3.add_module("5", nn.MaxPool2d(3, 2))
        self.slice3.add_module("6", nn.Conv2d(192, 384, 3, padding=1))
        self.slice3.add_module("7", nn.ReLU(inplace=False))
        self.slice4.add_module("8", nn.Conv2d(384, 256, 3, padding=1))
        self.slice4.add_module("9", nn.ReLU(inplace=False))
        self.slice5.add_module("10", nn.Conv2d(256, 256, 3, padding=1))
        self.slice5.add_module("11", nn.ReLU(inplace=False))

    def forward(self, x):
        outs = []
        for s in (self.slice1, self.slice2, self.slice3, self.slice4, self.slice5):
            x = s(x)
            outs.append(x)
        return outs


class _NetLinLayer(nn.Module):
    def __init__(self, chn_in: int):
        super().__init__()
        self.model = nn.Sequential(nn.Dropout(), nn.Conv2d(chn_in, 1, 1, stride=1, padding=0, bias=False))

    def forward(self, x):
        return self.model(x)


class LPIPSAlex(nn.Module):
    """``forward(in0, in1) -> [B,1,1,1]`` for images in [-1, 1]; eval mode, parameters frozen."""

    CHNS = (64, 192, 384, 256, 256)

    def __init__(self, state_dict: Optional[dict] = None, allow_partial: bool = False):
        super().__init__()
        self.scaling_layer = _ScalingLayer()
        self.net = _AlexFeatures()
        for i, c in enumerate(self.CHNS):
            setattr(self, f"lin{i}", _NetLinLayer(c))
        self.lins = [getattr(self, f"lin{i}") for i in range(5)]
        if state_dict is not None:
            # `lpips.LPIPS.state_dict()` also lists the lin layers a second time under `lins.N.*` (a ModuleList alias):
            # those duplicates are the only keys that may be ignored.  Anything else missing or unexpected would leave
            # random weights in place silently, so it raises unless the caller opts in with allow_partial=True
            sd = {k: v for k, v in state_dict.items() if not k.startswith("lins.")}
            res = self.load_state_dict(sd, strict=False)
            if (res.missing_keys or res.unexpected_keys) and not allow_partial:
                raise RuntimeError(f"LPIPSAlex: state dict does not match lpips.LPIPS(net='alex'): missing "
                                   f"{res.missing_keys[:6]}, unexpected {res.unexpected_keys[:6]} (allow_partial=True "
                                   f"loads what matches and leaves the rest random-initialised)")
            self.partial_keys = list(res.missing_keys)
        else:
            with torch.no_grad():           # LPIPS lin weights are non-negative
                for lin in self.lins:
                    lin.model[1].weight.abs_()
        self.eval().requires_grad_(False)

    @staticmethod
    def _normalize(x, eps: float = 1e-10):
        return x / (torch.sqrt(torch.sum(x ** 2, dim=1, keepdim=True)) + eps)

    accepts_unpooled = True       # forward(in0, in1) takes an in1 at 2x the resolution of in0 (see below)

    def _features_of_unpooled(self, x: torch.Tensor):
        """AlexNet features of AdaptiveAvgPool2d(H/2)(x) WITHOUT forming the pooled image: the 2 x 2 average folds into
        the first conv (11 x 11, stride 4, pad 2 on the pooled image == 22 x 22, stride 8, pad 4 on x with every tap
        duplicated 2 x 2 and divided by 4; the per-channel scaling layer commutes with the average).  The reference pools
        first (trainer_rgb.py:84-87); here the L2 term keeps its fused pool + MSE pass and LPIPS reads the 512^2 image."""
        c0 = self.net.slice1[0]
        w = c0.weight.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3) * 0.25
        y = F.relu(F.conv2d(self.scaling_layer(x), w, c0.bias, stride=8, padding=4))
        outs = [y]
        for s in (self.net.slice2, self.net.slice3, self.net.slice4, self.net.slice5):
            y = s(y)
            outs.append(y)
        return outs

    def forward(self, in0: torch.Tensor, in1: torch.Tensor) -> torch.Tensor:
        f0 = self.net(self.scaling_layer(in0))
        if in1.shape[-1] == 2 * in0.shape[-1] and in1.shape[-2] == 2 * in0.shape[-2]:
            f1 = self._features_of_unpooled(in1)
        else:
            f1 = self.net(self.scaling_layer(in1))
        val = 0
        for k in range(5):
            d = (self._normalize(f0[k]) - self._normalize(f1[k])) ** 2
            val = val + self.lins[k](d).mean(dim=(2, 3), keepdim=True)
        return val

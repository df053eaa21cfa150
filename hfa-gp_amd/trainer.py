"""Optimisation-step harness — counterpart of /root/reference/code/trainer_rgb.py:46-151 and
trainer_3dmm.py:20-122 (`Trainer.gen_update / sample / sample_bases / tune_generator / save / resume`).

Differences that are deliberate:
  * multi-GPU: one process per GPU, frames sharded by the caller; the gradients of the SHARED
    parameters (`bases`, `delta`, driver net, and the generator once it is being tuned) are summed with
    ONE flattened all-reduce per step over RCCL (`torch.distributed`, backend "nccl" on ROCm) and
    divided by the world size.  The reference wraps the module in DDP but its RGB trainer bypasses
    `DDP.forward`, so it never synchronises gradients at all; `north_star` asks for the all-reduce.
  * LPIPS(alex) weights cannot be obtained offline: `lpips=None` (default) trains on the L2 term only;
    a callable `lpips(real, fake) -> tensor` can be supplied.
  * no per-step `.item()` host sync; losses are returned as device tensors.
"""
from __future__ import annotations

import math
import os
from typing import Callable, Iterable, List, Optional

import torch
import torch.nn.functional as F
from torch import nn

from .cam_utils import create_cam2world_matrix, make_label, sample_camera_positions
from .headnerf import HeadNeRF_3DMM, HeadNeRF_final


def requires_grad(net: nn.Module, flag: bool = True) -> None:
    for p in net.parameters():
        p.requires_grad = flag


def allreduce_shared_grads(params: Iterable[torch.Tensor], world_size: int, group=None) -> int:
    """Sum-then-average the .grad of `params` over all ranks with ONE collective on one flat fp32 buffer
    (bases 50x7168 + delta 7168 = 1.46 MB; + driver net).  Returns the element count."""
    import torch.distributed as dist
    grads: List[torch.Tensor] = []
    for p in params:
        if p.requires_grad:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            grads.append(p.grad)
    if not grads:
        return 0
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world_size)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off: off + n].view_as(g))
        off += n
    return off


class Trainer(nn.Module):
    """`mode='rgb'` mirrors trainer_rgb.Trainer (image-driven), `mode='3dmm'` trainer_3dmm.Trainer."""

    def __init__(self, args, device, rank: int = 0, world_size: int = 1, mode: str = "rgb",
                 lpips: Optional[Callable] = None, gen: Optional[nn.Module] = None):
        super().__init__()
        self.args, self.device, self.rank, self.world_size, self.mode = args, device, rank, world_size, mode
        self.batch_size = args.batch_size
        if gen is None:
            cls = HeadNeRF_final if mode == "rgb" else HeadNeRF_3DMM
            gen = cls(args, args.size, device, args.latent_dim_style, args.latent_dim_shape,
                      getattr(args, "run_id", "nerface2"), getattr(args, "emb_dir", "./PTI/embeddings/"))
        self.gen = gen.to(device)
        # Adam over ALL parameters, THEN freeze the generator — same order as trainer_rgb.py:58-60, so
        # that tune_generator() starts updating the generator without rebuilding the optimiser.
        self.g_optim = torch.optim.Adam(self.gen.parameters(), lr=args.lr)
        requires_grad(self.gen.generator, False)
        self.lpips_loss = lpips
        self.face_pool = nn.AdaptiveAvgPool2d((args.size, args.size))
        if world_size > 1:
            self.broadcast_parameters()

    # ------------------------------------------------------------------ distributed
    def broadcast_parameters(self) -> None:
        import torch.distributed as dist
        for t in list(self.gen.parameters()) + list(self.gen.buffers()):
            dist.broadcast(t.data, src=0)

    def shared_parameters(self):
        return [p for p in self.gen.parameters() if p.requires_grad]

    # ------------------------------------------------------------------ reference API
    def l2_loss(self, real_images, generated_images):
        return F.mse_loss(real_images, generated_images, reduction="mean")

    def tune_generator(self):
        requires_grad(self.gen.generator, True)

    def gen_update(self, real_image, label, params=None, person_2=False):
        self.gen.train()
        self.g_optim.zero_grad()
        if self.mode == "rgb":
            weights = self.gen.get_weights(real_image)
            latent = self.gen.get_latent(weights, person_2)
            generated = self.gen.get_image(latent, label)
        else:
            generated = self.gen(params, label, person_2)
        generated = self.face_pool(generated)
        l2 = self.l2_loss(real_image, generated)
        if self.lpips_loss is not None:
            lp = torch.squeeze(self.lpips_loss(real_image, generated)).mean()
        else:
            lp = torch.zeros((), device=l2.device)
        (l2 + lp).backward()
        if self.world_size > 1:
            allreduce_shared_grads(self.shared_parameters(), self.world_size)
        self.g_optim.step()
        return l2.detach(), lp.detach(), generated.detach()

    def sample(self, real_image, label, params=None, person_2=False):
        with torch.no_grad():
            self.gen.eval()
            if self.mode == "rgb":
                return self.gen(real_image, label, person_2)
            return self.gen(params, label, person_2)

    def frontal_label(self, r: float = 2.7) -> torch.Tensor:
        pts, _, _ = sample_camera_positions(device=self.device, n=1, r=r, horizontal_mean=0.5 * math.pi,
                                            vertical_mean=0.5 * math.pi, mode=None)
        return make_label(create_cam2world_matrix(-pts, pts, device=self.device))

    def sample_bases(self, person_2=False, scale: float = 10.0):
        """One render per basis vector (alpha = scale * e_i).  The label tensor is re-used across calls,
        so — exactly as in the reference (trainer_rgb.py:113-125 + headnerf.py:132) — odd and even bases
        see flipped / un-flipped cameras."""
        imgs = []
        with torch.no_grad():
            label = self.frontal_label()
            self.gen.eval()
            k = self.args.latent_dim_shape
            for i in range(k):
                w = torch.zeros(1, k, device=self.device)
                w[0, i] = scale
                imgs.append(self.gen.get_image(self.gen.get_latent(w, person_2), label))
        return imgs

    def resume(self, resume_ckpt: str) -> int:
        ckpt = torch.load(resume_ckpt, map_location=self.device, weights_only=False)
        start_iter = int(os.path.splitext(os.path.basename(resume_ckpt))[0])
        self.gen.load_state_dict(ckpt["gen"])
        self.g_optim.load_state_dict(ckpt["g_optim"])
        return start_iter

    def save(self, idx: int, checkpoint_path: str) -> str:
        path = f"{checkpoint_path}/{str(idx).zfill(6)}.pt"
        torch.save({"gen": self.gen.state_dict(), "g_optim": self.g_optim.state_dict(), "args": self.args}, path)
        return path

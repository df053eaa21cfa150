"""Optimisation-step harness — counterpart of /root/reference/code/trainer_rgb.py:46-151 and
trainer_3dmm.py:20-122 (`Trainer.gen_update / sample / sample_bases / tune_generator / save / resume`).

Differences that are deliberate:
  * multi-GPU: one process per GPU; `fit_frames` shards the frame set in contiguous blocks over the ranks
    (`shard_range` / `epoch_batches`, ragged tails included); the gradients of the SHARED parameters (`bases`,
    `delta`, driver net, and the generator once it is being tuned) live in ONE persistent flat fp32 buffer
    (`FlatGrads`: the `.grad` tensors are its slices) that is all-reduced in place per step over RCCL
    (`torch.distributed`, backend "nccl" on ROCm) on the launch stream and averaged over the world size.  The reference wraps the module in DDP but its RGB trainer bypasses
    `DDP.forward`, so it never synchronises gradients at all; `north_star` asks for the all-reduce.
  * LPIPS(alex) weights cannot be obtained offline: a callable `lpips(real, fake) -> tensor` can be supplied
    (`lpips_alex.LPIPSAlex` has the lpips package's architecture and key names); `lpips=None` trains on the L2
    term only and WARNS, `lpips='none'` does so silently.
  * no per-step `.item()` host sync; losses are returned as device tensors.
"""
from __future__ import annotations

import math
import os
import warnings
from typing import Callable, Iterable, List, Optional, Tuple, Union

import torch
import torch.nn.functional as F
from torch import nn

from .cam_utils import create_cam2world_matrix, make_label, sample_camera_positions
from .headnerf import AudioAttNet, AudioNet, HeadNeRF_3DMM, HeadNeRF_Audio, HeadNeRF_final


def pooled_l2(face_pool: nn.Module, real_image: torch.Tensor, generated: torch.Tensor, need_pooled_grad: bool):
    """(l2, pooled image) of the reference's  generated = face_pool(generated); l2 = MSE(real, generated)
    (trainer_rgb.py:84-86).  On the MI355X with an integer pooling factor and no LPIPS term this is the fused
    pass of ops.pool_mse (one read of the image forward, one write backward); otherwise PyTorch-ROCm ops."""
    if (generated.is_cuda and not need_pooled_grad and real_image.dtype == torch.float32
            and real_image.shape[:2] == generated.shape[:2]
            and generated.shape[-2] % real_image.shape[-2] == 0 and generated.shape[-1] % real_image.shape[-1] == 0
            and generated.shape[-2] // real_image.shape[-2] == generated.shape[-1] // real_image.shape[-1]):
        from . import ops
        return ops.pool_mse(generated.contiguous(), real_image.contiguous())
    pooled = face_pool(generated)
    return F.mse_loss(real_image, pooled, reduction="mean"), pooled


class MultiTensorAdam(torch.optim.Adam):
    """torch.optim.Adam (same constructor, same `state_dict` layout as its fused implementation: per parameter a device float
    `step`, `exp_avg`, `exp_avg_sq`) whose `step()` is ONE launch of the library's multi-tensor kernel (hfagp_adam_step) over
    device-resident pointer tables — PyTorch's fused Adam is 19 launches at 1.9 TB/s for the 37.7 M parameters of the tuned step
    (0.55 ms); the update is HBM-bound (28 bytes per parameter).  Same rule, bias corrections in double as torch forms them;
    parameters whose `.grad` is None are skipped like torch's.  Options the reference never sets (weight decay, amsgrad, maximize,
    a tensor lr) take torch's own path."""

    def __init__(self, params, **kw):
        super().__init__(params, **kw)
        self._tables = {}

    @torch.no_grad()
    def step(self, closure=None):
        # options the kernel does not implement: torch's own step — decided BEFORE the closure runs, so that its loss is torch's to return
        for group in self.param_groups:
            if (group["weight_decay"] != 0 or group["amsgrad"] or group["maximize"] or isinstance(group["lr"], torch.Tensor)
                    or group.get("differentiable", False)):
                return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from . import ops
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            for p in params:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                elif st["step"].device != p.device or st["step"].dtype != torch.float32:
                    # (a checkpoint written by the non-fused torch.optim.Adam keeps `step` on the host)
                    st["step"] = st["step"].to(device=p.device, dtype=torch.float32)
            key = (gi,) + tuple((id(p), p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(),
                                 self.state[p]["exp_avg_sq"].data_ptr(), self.state[p]["step"].data_ptr()) for p in params)
            tab = self._tables.get(key)
            if tab is None:
                if len(self._tables) > 8:
                    self._tables.clear()
                tab = self._tables[key] = ops.AdamTables(params, [p.grad for p in params], [self.state[p]["exp_avg"] for p in params],
                                                         [self.state[p]["exp_avg_sq"] for p in params], [self.state[p]["step"] for p in params])
            b1, b2 = group["betas"]
            ops.adam_step(tab, float(group["lr"]), float(b1), float(b2), float(group["eps"]))
            # the kernel writes the parameters through raw pointers: tell autograd (and every cache keyed on `_version`: the
            # generator's weight images, the encoder trunk's, the latent basis' Q) that they changed, as an in-place torch op would
            _bump_versions(params)
        return loss


def _bump_versions(params) -> None:
    inc = getattr(torch.autograd.graph, "increment_version", None)
    if inc is not None:
        try:
            inc(params)                     # (an iterable of tensors since torch 2.5)
            return
        except TypeError:
            for p in params:
                inc(p)
            return
    for p in params:                        # older torch: an in-place no-op that goes through the dispatcher
        p.add_(0)


def _adam(params, **kw) -> torch.optim.Adam:
    """torch.optim.Adam as the reference builds it (trainer_rgb.py:58); on CUDA/ROCm fp32 parameters `MultiTensorAdam` (one launch of
    the library's kernel per step; HFAGP_TORCH_ADAM=1: PyTorch's fused implementation).  Same state_dict layout, same update rule."""
    params = list(params)
    if params and all(p.is_cuda and p.dtype == torch.float32 for p in params):
        if os.environ.get("HFAGP_TORCH_ADAM") != "1":
            return MultiTensorAdam(params, **kw)
        try:
            return torch.optim.Adam(params, fused=True, **kw)
        except (RuntimeError, TypeError):
            pass
    return torch.optim.Adam(params, **kw)


def requires_grad(net: nn.Module, flag: bool = True) -> None:
    for p in net.parameters():
        p.requires_grad = flag


def allreduce_shared_grads(params: Iterable[torch.Tensor], world_size: int, group=None) -> int:
    """Sum-then-average the .grad of `params` over all ranks with ONE collective on one flat fp32 buffer
    (bases 50x7168 + delta 7168 = 1.46 MB; + driver net).  Returns the element count.
    Stand-alone form (builds the flat buffer per call); the trainers use `FlatGrads`, whose buffer is persistent."""
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


class FlatGrads:
    """ONE persistent flat fp32 buffer whose slices ARE the `.grad` tensors of the shared parameters (SURVEY.md
    §5.8 / §8e): autograd accumulates straight into it, `zero()` is one memset, and the all-reduce runs on the
    buffer itself on the launch stream right after the backward pass — no `cat`, no copy back.

    Buckets: the parameters are laid out in the order given; `bucket_bytes` cuts the buffer into contiguous
    buckets that are reduced by separate collectives (latency-bound basis/driver gradients: one bucket; the 123 MB of
    a generator being tuned: one bucket per ~32 MB so the ring pipeline of RCCL stays busy, 7 x 153 GB/s xGMI links).
    `rebuild()` must be called when the set of parameters that require grad changes (`tune_generator`)."""

    def __init__(self, params: Iterable[torch.Tensor], bucket_bytes: int = 32 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.bucket_bytes = bucket_bytes
        self.numel = sum(p.numel() for p in self.params)
        self.flat: Optional[torch.Tensor] = None
        self.buckets: List[Tuple[int, int]] = []
        # every slice starts on a 16-byte boundary (a 0-d noise strength in front of a conv weight left the weight's
        # .grad 4-byte aligned, and the reducers / the Adam kernel access it with 16-byte loads): `offsets[i]` is where parameter
        # i's slice starts, the padding floats in between stay zero and travel with the collectives
        self.offsets: List[int] = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.size = off
        if self.params:
            p0 = self.params[0]
            self.flat = torch.zeros(self.size, device=p0.device, dtype=torch.float32)
            b0 = 0
            for p, o in zip(self.params, self.offsets):
                n = p.numel()
                p.grad = self.flat[o: o + n].view_as(p)
                off = o + (n + 3) // 4 * 4
                if (off - b0) * 4 >= bucket_bytes:
                    self.buckets.append((b0, off))
                    b0 = off
            if off > b0:
                self.buckets.append((b0, off))

    def owns(self, params: Iterable[torch.Tensor]) -> bool:
        """True while every parameter's .grad still is this buffer's slice (an optimiser `zero_grad(set_to_none=True)`
        or a changed `requires_grad` set breaks the aliasing)."""
        want = [p for p in params if p.requires_grad]
        if len(want) != len(self.params) or any(a is not b for a, b in zip(want, self.params)):
            return False
        base = self.flat.data_ptr() if self.flat is not None else 0
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != base + 4 * off:
                return False
        return True

    def zero(self) -> None:
        if self.flat is not None:
            self.flat.zero_()

    def allreduce_mean(self, world_size: int, group=None) -> int:
        """In-place mean over the ranks, one collective per bucket, enqueued on the current (launch) stream."""
        if self.flat is None or world_size <= 1:
            return 0
        import torch.distributed as dist
        avg = getattr(dist.ReduceOp, "AVG", None) if dist.get_backend(group) == "nccl" else None
        for lo, hi in self.buckets:
            seg = self.flat[lo:hi]
            if avg is not None:
                dist.all_reduce(seg, op=avg, group=group)          # RCCL averages in the collective: no second pass
            else:
                dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=group)
        if avg is None:
            self.flat.div_(world_size)
        return self.numel


class BucketedAllReduce:
    """Overlaps the all-reduce of `FlatGrads` buckets with the rest of the backward pass (when the
    generator is tuned the shared gradients are 123 MB — bandwidth-bound on xGMI — and should travel while the
    backward pass is still running).  A bucket becomes READY when the last of its parameters has its final gradient:
      * generator parameters: `SynthesisFn.backward` hands them to `generator._grad_sink` block by block
        (autograd.py `release_ready`), in the order super-resolution -> decoder -> backbone 256 ... 4 -> affine layers;
      * everything else (basis, driver net): `register_post_accumulate_grad_hook`;
      * parameters the caller declares ABSENT for this step (`begin_step(absent=...)`: the other identity's basis, nets
        that are not on this step's path) count as ready from the start.
    Collectives are STARTED IN BUCKET INDEX ORDER ONLY — bucket b goes out (async, on RCCL's own stream, ordered after
    the launch stream) once it is ready AND buckets 0 .. b-1 are out — so every rank issues the same collective
    sequence whatever order its own backward pass finished things in, including a rank whose batch is empty (ragged
    shard tail: it runs no backward pass and `finish()` starts all buckets, in index order).  The flat buffer is laid out
    in readiness order (`_readiness_order`), so in-order launching costs no overlap.
    `finish()` starts whatever is left, waits, and averages."""

    def __init__(self, flat: FlatGrads, world_size: int, group=None):
        import torch.distributed as dist
        self.flat, self.world, self.group = flat, world_size, group
        self.avg = getattr(dist.ReduceOp, "AVG", None) if dist.get_backend(group) == "nccl" else None
        self.bucket_of = {}
        self.size = [0] * len(flat.buckets)
        b = 0
        for p, off in zip(flat.params, flat.offsets):
            while off >= flat.buckets[b][1]:
                b += 1
            self.bucket_of[id(p)] = b
            self.size[b] += 1
        self.active = False
        self.last_order: List[int] = []
        self.generator_ids = set()     # ids of generator parameters (`_StepScope`): their hook also fires without a write
        self.reset()

    def reset(self):
        """Forget the state of an interrupted step: outstanding collectives are waited for (their buffers are about to be
        zeroed), the counters start over."""
        for w in getattr(self, "works", []):
            if w is not None:
                w.wait()
        self.left = list(self.size)
        self.works = [None] * len(self.size)
        self.seen = set()
        self.declared_absent = set()   # ids the caller declared gradient-free for this step (begin_step)
        self.next = 0                  # the next bucket to go out: buckets < next are in flight or done
        self.order = []                # buckets in the order their collectives were started (tests / diagnostics)

    def begin_step(self, absent: Iterable[torch.Tensor] = ()):
        """Start of a step (after `FlatGrads.zero()`): fresh counters; `absent` parameters receive no gradient in this
        step and must not hold their bucket back."""
        self.reset()
        self.active = True
        absent = list(absent)
        self.declared_absent = {id(p) for p in absent}
        for p in absent:
            self._count(p)
        self._advance()

    def _launch(self, b: int):
        import torch.distributed as dist
        lo, hi = self.flat.buckets[b]
        op = self.avg if self.avg is not None else dist.ReduceOp.SUM
        self.works[b] = dist.all_reduce(self.flat.flat[lo:hi], op=op, group=self.group, async_op=True)
        self.order.append(b)

    def _advance(self):
        while self.next < len(self.size) and self.left[self.next] == 0:
            self._launch(self.next)
            self.next += 1

    def _count(self, p: torch.Tensor) -> Optional[int]:
        b = self.bucket_of.get(id(p))
        if b is None or id(p) in self.seen:
            return b
        self.seen.add(id(p))
        self.left[b] -= 1
        return b

    def on_grad(self, p: torch.Tensor):
        """Post-accumulate notification (autograd hook).  PyTorch also fires it for a parameter whose gradient a custom
        Function returned as None — every generator parameter that went through the sink is notified a second time WITHOUT a
        write — so a notification for a parameter that is already counted is ignored; real writes into a bucket in flight are
        caught where they happen (`sink`)."""
        if not self.active:
            return                     # a backward pass outside gen_update (autograd.grad, a sample): nothing to overlap
        b = self.bucket_of.get(id(p))
        if b is None:
            return
        if id(p) in self.seen:
            # generator parameters pass here a second time without a write (see above).  Anything ELSE that is already counted
            # was declared absent for this step (`begin_step(absent=...)`) and has received a gradient after all: its bucket may
            # be in flight with the stale zeros, the ranks would diverge silently and `step_skipping` would hide the gradient
            # from Adam — refuse
            if id(p) in self.declared_absent and id(p) not in self.generator_ids:
                raise RuntimeError("BucketedAllReduce: a parameter that was declared ABSENT for this step received a gradient "
                                   f"(shape {tuple(p.shape)}, bucket {b}, collective {'in flight' if self.works[b] is not None else 'pending'}); "
                                   "fix Trainer.absent_parameters — absent parameters are skipped by Adam and their bucket "
                                   "does not wait for them")
            return
        self._count(p)
        self._advance()

    def sink(self, p: torch.Tensor, g: torch.Tensor):
        """generator._grad_sink: accumulate into the flat slice (g None: the backward pass already did, in place), then count
        the parameter as ready."""
        b = self.bucket_of.get(id(p))
        if self.active and b is not None and self.works[b] is not None:
            raise RuntimeError("BucketedAllReduce: a generator gradient arrived for a parameter whose bucket is already being "
                               "all-reduced (the parameter was declared absent, or its gradient is produced twice in one "
                               "step — e.g. two synthesis calls in one graph); sum the losses into ONE backward pass and "
                               "do not list used parameters as absent")
        if g is not None:
            p.grad.add_(g.view_as(p.grad))
        self.on_grad(p)

    def finish(self):
        self.launched_early = self.next        # buckets that went out from inside the backward pass (diagnostics / tests)
        for b in range(len(self.size)):
            self.left[b] = 0
        self._advance()
        for w in self.works:
            w.wait()
        if self.avg is None:
            self.flat.flat.div_(self.world)
        n = self.flat.numel
        self.last_order = self.order
        self.works = [None] * len(self.size)
        self.active = False
        return n


def shard_range(n_frames: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous-block frame shard [lo, hi) of rank `rank` (2000 frames over 8 ranks →
    rank r owns [250 r, 250 (r+1)); contiguous blocks keep audio smoothing windows local, §8e).  When
    `n_frames % world_size != 0` the first `n_frames % world_size` ranks own one frame more (ragged tail)."""
    base, extra = divmod(n_frames, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def epoch_batches(n_frames: int, rank: int, world_size: int, batch: int):
    """One pass over a contiguous-block sharded frame set: yields (frame indices of THIS rank [b_r], weight) per step;
    every rank yields the SAME number of steps (= ceil(largest shard / batch)) so the per-step collective matches.
    A rank whose shard is exhausted yields an empty index tensor (it still joins the all-reduce with zero gradients).
    `weight` = b_r * world / (sum over ranks of b) rescales the rank's mean loss so that the all-reduce MEAN equals the
    gradient of the mean over all frames of the step, also when the last batches are ragged."""
    shards = [shard_range(n_frames, r, world_size) for r in range(world_size)]
    steps = max(-(-(hi - lo) // batch) for lo, hi in shards) if n_frames > 0 else 0
    for s in range(steps):
        counts = [max(0, min(batch, hi - lo - s * batch)) for lo, hi in shards]
        lo, _ = shards[rank]
        idx = torch.arange(lo + s * batch, lo + s * batch + counts[rank])
        yield idx, counts[rank] * world_size / max(sum(counts), 1)


_BASIS_NAMES = ("bases", "delta", "bases_2", "delta_2")


def _readiness_order(*modules: nn.Module) -> List[torch.Tensor]:
    """Trainable parameters of `modules` ordered by when the backward pass finishes their gradients (see
    Trainer.shared_parameters)."""
    first, affine, rest, idle, seen = [], [], [], [], set()
    for m in modules:
        gen = getattr(m, "generator", None)
        if gen is not None:
            named = dict(gen.named_parameters())
            sr = [n for n in named if n.startswith("superresolution.block1.")] + \
                 [n for n in named if n.startswith("superresolution.block0.")]
            dec = [n for n in named if n.startswith("decoder.")]
            bb = [n for n in named if n.startswith("backbone.synthesis.")]
            res = sorted({int(n.split(".")[2][1:]) for n in bb}, reverse=True)
            bb = [n for r in res for n in bb if n.split(".")[2] == f"b{r}"]
            other = [n for n in named if n not in set(sr) | set(dec) | set(bb)]
            cfg = getattr(gen, "cfg", None)

            idle_ids = {id(p) for p in _generator_idle_parameters(gen)}   # no gradient ever arrives: they must not hold a bucket back
            for n in sr + dec + bb + other:
                p = named[n]
                if p.requires_grad and id(p) not in seen:
                    seen.add(id(p))
                    (idle if id(p) in idle_ids else affine if ".affine." in n else first).append(p)
        # everything that is not the generator: the backward pass reaches the latent basis first (QR adjoint, right after
        # d ws), then walks the driver net from its LAST layer to its first — reverse registration order
        own = [(n, p) for n, p in m.named_parameters() if p.requires_grad and id(p) not in seen]
        basis = [(n, p) for n, p in own if n.split(".")[-1] in _BASIS_NAMES]
        driver = [(n, p) for n, p in own if n.split(".")[-1] not in _BASIS_NAMES]
        for n, p in basis + driver[::-1]:
            seen.add(id(p))
            rest.append(p)
    return first + affine + rest + idle


def _install_grad_hooks(trainer, modules) -> None:
    """Route final gradients of the non-generator parameters to the trainer's CURRENT BucketedAllReduce (looked up at call
    time: the bucketer is rebuilt when the set of trainable parameters changes).  Installed once per parameter; the hook
    does nothing outside a step (`BucketedAllReduce.active`)."""
    def current():
        return getattr(trainer, "_bucketer", None)
    for m in modules:
        for p in m.parameters():
            if getattr(p, "_hfagp_hooked", False) or not p.requires_grad:
                continue                       # (frozen now: hooked when it becomes trainable and the buffer is rebuilt)
            p._hfagp_hooked = True

            def hook(param, _cur=current):
                b = _cur()
                if b is not None:
                    b.on_grad(param)
            p.register_post_accumulate_grad_hook(hook)


class _StepScope:
    """The part of a step during which gradients flow into the bucketer: `begin_step`, the generator's gradient sink
    installed (generator parameters are then released block by block from inside `SynthesisFn.backward`), and on the
    way out — also when forward or backward raised — the sink removed and the bucketer left inactive, so that a later
    backward pass outside `gen_update` (autograd.grad, a second trainer sharing the generator) sees plain autograd."""

    def __init__(self, bucketer: Optional[BucketedAllReduce], generators, absent):
        self.b, self.gens, self.absent = bucketer, [g for g in generators if g is not None], absent

    def __enter__(self):
        for g in self.gens:
            # SynthesisFn.backward adds the generator's gradients into their .grad slices itself (HFAGP_GRAD_INPLACE=0: autograd does)
            g._grad_inplace = os.environ.get("HFAGP_GRAD_INPLACE", "1") != "0"
        if self.b is not None:
            if not self.b.generator_ids:
                self.b.generator_ids = {id(p) for g in self.gens for p in g.parameters()}
            self.b.begin_step(self.absent)
            for g in self.gens:
                g._grad_sink = self.b.sink
        return self.b

    def __exit__(self, exc_type, exc, tb):
        for g in self.gens:
            g._grad_inplace = False
        if self.b is not None:
            for g in self.gens:
                g._grad_sink = None
            if exc_type is not None:
                self.b.active = False
        return False


def _generator_idle_parameters(gen) -> List[torch.Tensor]:
    """Trainable generator parameters that are never on the synthesis path (mapping network; noise strengths when the
    noise mode is 'none'): they receive no gradient, ever."""
    out = []
    cfg = getattr(gen, "cfg", None)
    for n, p in gen.named_parameters():
        if not p.requires_grad:
            continue
        if n.startswith("backbone.mapping."):
            out.append(p)
        elif n.endswith(".noise_strength") and cfg is not None:
            mode = cfg.sr_noise_mode if n.startswith("superresolution.") else cfg.backbone_noise_mode
            if mode == "none":
                out.append(p)
    return out


def step_skipping(optimizers, absent: Iterable[torch.Tensor]) -> None:
    """`optimizer.step()` with the reference's `zero_grad()` semantics (set_to_none: trainer_rgb.py:75) on a PERSISTENT
    gradient buffer: parameters that received no gradient this step (`absent`) must not take an Adam step — no moment
    decay, no step-count advance — although their `.grad` slice exists and is zero.  Their `.grad` is hidden for the
    duration of the step and restored (still the FlatGrads slice) afterwards."""
    hidden = [(p, p.grad) for p in absent if p.grad is not None]
    for p, _ in hidden:
        p.grad = None
    try:
        for opt in optimizers:
            opt.step()
    finally:
        for p, g in hidden:
            p.grad = g


class Trainer(nn.Module):
    """`mode='rgb'` mirrors trainer_rgb.Trainer (image-driven: `gen_update(real, label, person_2)` → (l2, lpips, img),
    optimiser / checkpoint key `g_optim`), `mode='3dmm'` mirrors trainer_3dmm.Trainer (`gen_update(real, label, params,
    person_2)` → (l2_3dmm = zeros(1), l2, lpips, img), optimiser / checkpoint key `w_optim`, `sample_bases` with
    alpha = 5 e_i instead of 10 e_i)."""

    def __init__(self, args, device, rank: int = 0, world_size: int = 1, mode: str = "rgb",
                 lpips: Union[Callable, str, None] = None, gen: Optional[nn.Module] = None):
        super().__init__()
        if mode not in ("rgb", "3dmm"):
            raise ValueError(f"mode must be 'rgb' or '3dmm', got {mode!r}")
        self.args, self.device, self.rank, self.world_size, self.mode = args, device, rank, world_size, mode
        self.batch_size = args.batch_size
        if gen is None:
            cls = HeadNeRF_final if mode == "rgb" else HeadNeRF_3DMM
            gen = cls(args, args.size, device, args.latent_dim_style, args.latent_dim_shape,
                      getattr(args, "run_id", "nerface2"), getattr(args, "emb_dir", "./PTI/embeddings/"))
        self.gen = gen.to(device)
        # Adam over ALL parameters, THEN freeze the generator — same order as trainer_rgb.py:58-60 /
        # trainer_3dmm.py:33-35, so that tune_generator() starts updating the generator without rebuilding the optimiser.
        self.optim_key = "g_optim" if mode == "rgb" else "w_optim"
        setattr(self, self.optim_key, _adam(self.gen.parameters(), lr=args.lr))
        requires_grad(self.gen.generator, False)
        if lpips is None:
            # the reference objective is ALWAYS l2 + LPIPS(alex) (trainer_rgb.py:62,86-91); its weights cannot be
            # obtained offline, so an L2-only step is legal but must not be silent
            warnings.warn("Trainer: no LPIPS module given — optimising the L2 term only (the reference trains on "
                          "l2 + LPIPS(alex)); pass lpips=LPIPSAlex(state_dict) or lpips='none' to silence this",
                          stacklevel=2)
        self.lpips_loss = None if (lpips is None or (isinstance(lpips, str) and lpips == "none")) else lpips
        self.face_pool = nn.AdaptiveAvgPool2d((args.size, args.size))
        self.timing: Optional[dict] = None   # bench.py: {'fwd': [(ev0, ev1)], 'bwd': ..., 'allreduce': ..., 'optim': ...}
        if world_size > 1:
            self.broadcast_parameters()
        self._flat: Optional[FlatGrads] = None

    @property
    def optimizer(self) -> torch.optim.Optimizer:
        return getattr(self, self.optim_key)

    @optimizer.setter
    def optimizer(self, opt: torch.optim.Optimizer) -> None:
        setattr(self, self.optim_key, opt)

    # ------------------------------------------------------------------ distributed
    def broadcast_parameters(self) -> None:
        """Rank 0's parameters and buffers to every rank, in place on the tensors themselves (not `.data`: that
        would leave `_version` unchanged and the generator's weight-image caches stale)."""
        import torch.distributed as dist
        with torch.no_grad():
            for t in list(self.gen.parameters()) + list(self.gen.buffers()):
                dist.broadcast(t, src=0)
        inv = getattr(self.gen.generator, "invalidate_caches", None)
        if inv is not None:
            inv()

    def shared_parameters(self):
        """Parameters whose gradients are shared by the ranks, in FLAT-BUFFER order = the order in which the backward
        pass finishes them, so that contiguous buckets complete early: generator tensors as `SynthesisFn.backward`
        releases them (super-resolution, decoder, backbone 256 ... 4; the affine layers, whose gradients come out of the
        batched style adjoint at the very end, last), then basis / driver parameters (finished after the generator)."""
        return _readiness_order(self.gen)

    def flat_grads(self) -> FlatGrads:
        """The persistent gradient buffer of the parameters that currently require grad (rebuilt when that set
        changes, e.g. after `tune_generator`, or when something replaced a .grad tensor)."""
        shared = self.shared_parameters()
        if self._flat is None or not self._flat.owns(shared):
            self._flat = FlatGrads(shared, getattr(self, "bucket_bytes", 32 << 20))
            self._bucketer = None
        return self._flat

    def _overlap(self, flat: FlatGrads) -> Optional[BucketedAllReduce]:
        """The bucketed, backward-overlapped all-reduce of `flat` (world_size > 1 only)."""
        # (force_collective: tests run the multi-rank code path in a 1-rank process group)
        if (self.world_size <= 1 and not getattr(self, "force_collective", False)) or flat.flat is None:
            return None
        if getattr(self, "_bucketer", None) is None:
            self._bucketer = BucketedAllReduce(flat, self.world_size)
            _install_grad_hooks(self, [self.gen])
        return self._bucketer

    def absent_parameters(self, person_2: bool = False) -> List[torch.Tensor]:
        """Trainable parameters that receive NO gradient in a step for identity `person_2` on every rank: the other
        identity's basis / mean (headnerf.py:60-69,86-90), the Encoder's pose head (its output is not in the loss,
        trainer_rgb.py:77-91), generator parameters off the synthesis path.  The reference's `zero_grad()` leaves their
        `.grad` None, so Adam skips them (`step_skipping`); the bucketer counts them as ready (`begin_step`)."""
        used = {id(t) for t in self.gen._select(person_2)}
        out = []
        for n, p in self.gen.named_parameters():
            if not p.requires_grad or n.startswith("generator."):
                continue
            if n in _BASIS_NAMES and id(p) not in used:
                out.append(p)
            elif n.startswith("encoder.pose."):
                out.append(p)
        return out + _generator_idle_parameters(self.gen.generator)

    # ------------------------------------------------------------------ reference API
    def l2_loss(self, real_images, generated_images):
        return F.mse_loss(real_images, generated_images, reduction="mean")

    def tune_generator(self):
        requires_grad(self.gen.generator, True)

    def _mark(self):
        """HIP event on the current stream when bench.py asked for the phase breakdown of the step."""
        if self.timing is None:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def _span(self, key, e0, e1):
        if self.timing is not None:
            self.timing.setdefault(key, []).append((e0, e1))

    def gen_update(self, real_image, label, params=None, person_2=False, loss_weight: float = 1.0, *,
                   u_strat: Optional[torch.Tensor] = None, u_imp: Optional[torch.Tensor] = None):
        """One optimisation step.  rgb mode: `gen_update(real, label, person_2=...)` → (l2, lpips, image);
        3dmm mode: `gen_update(real, label, params, person_2)` → (l2_3dmm, l2, lpips, image) — the reference's
        signatures and return arity (trainer_rgb.py:73-98, trainer_3dmm.py:43-67; train_3dmm.py:128 unpacks four).
        `loss_weight` rescales this rank's loss before the all-reduce mean (ragged frame shards, `epoch_batches`);
        an EMPTY batch skips the forward/backward and contributes zero gradients to the collective.
        `u_strat` / `u_imp` (keyword-only, TEST HOOK): the renderer's uniforms for this step instead of fresh draws
        (`_LatentBasis._synthesis`) — parity tests and the exact multi-rank equalities need two steps to sample alike."""
        if self.mode == "rgb" and isinstance(params, bool):          # reference call form gen_update(real, label, person_2)
            params, person_2 = None, params
        self.gen.train()
        flat = self.flat_grads()
        bucketer = self._overlap(flat)
        if bucketer is not None:
            bucketer.reset()           # (waits for collectives an interrupted step left behind before their buffer is zeroed)
        flat.zero()
        absent = self.absent_parameters(person_2)
        t0 = self._mark()
        empty = real_image.shape[0] == 0
        with _StepScope(bucketer, [self.gen.generator], absent):
            if empty:
                l2 = lp = torch.zeros((), device=self.device)
                generated = real_image
                t1 = t2 = self._mark()
            else:
                if self.mode == "rgb":
                    weights = self.gen.get_weights(real_image)
                    if isinstance(weights, tuple):
                        weights = weights[0]
                    latent = self.gen.get_latent(weights, person_2)
                    generated = self.gen.get_image(latent, label, u_strat=u_strat, u_imp=u_imp)
                else:
                    generated = self.gen(params, label, person_2, u_strat=u_strat, u_imp=u_imp)
                full = generated
                # LPIPSAlex folds a 2 x 2 pool into its first conv: the L2 term then keeps the fused pool + MSE pass and no
                # pooled image with a gradient is needed (lpips_alex.LPIPSAlex._features_of_unpooled)
                fold = (self.lpips_loss is not None and getattr(self.lpips_loss, "accepts_unpooled", False)
                        and full.shape[-1] == 2 * real_image.shape[-1] and full.shape[-2] == 2 * real_image.shape[-2])
                l2, generated = pooled_l2(self.face_pool, real_image, generated, self.lpips_loss is not None and not fold)
                if self.lpips_loss is not None:
                    lp = torch.squeeze(self.lpips_loss(real_image, full if fold else generated)).mean()
                else:
                    lp = torch.zeros((), device=l2.device)
                t1 = self._mark()
                g_loss = l2 + lp
                if loss_weight != 1.0:
                    g_loss = g_loss * loss_weight
                g_loss.backward()
                t2 = self._mark()
            if bucketer is not None:
                bucketer.finish()      # (most buckets are already in flight: started from inside the backward pass)
        t3 = self._mark()
        step_skipping([self.optimizer], absent)
        t4 = self._mark()
        self._span("fwd", t0, t1), self._span("bwd", t1, t2), self._span("allreduce", t2, t3), self._span("optim", t3, t4)
        if self.mode == "3dmm":
            return torch.zeros(1, device=self.device), l2.detach(), lp.detach(), generated.detach()
        return l2.detach(), lp.detach(), generated.detach()

    def sample(self, real_image, label, params=None, person_2=False):
        if self.mode == "rgb" and isinstance(params, bool):
            params, person_2 = None, params
        with torch.no_grad():
            self.gen.eval()
            if self.mode == "rgb":
                return self.gen(real_image, label, person_2)
            return self.gen(params, label, person_2)

    def frontal_label(self, r: float = 2.7) -> torch.Tensor:
        pts, _, _ = sample_camera_positions(device=self.device, n=1, r=r, horizontal_mean=0.5 * math.pi,
                                            vertical_mean=0.5 * math.pi, mode=None)
        return make_label(create_cam2world_matrix(-pts, pts, device=self.device))

    def sample_bases(self, person_2=False, scale: Optional[float] = None):
        """One render per basis vector (alpha = scale * e_i; 10 in trainer_rgb.py:120, 5 in trainer_3dmm.py:90).  The
        label tensor is re-used across calls, so — exactly as in the reference (trainer_rgb.py:113-125 +
        headnerf.py:132) — odd and even bases see flipped / un-flipped cameras."""
        if scale is None:
            scale = 10.0 if self.mode == "rgb" else 5.0
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
        # the reference's own key for this mode first; the other trainer's key is accepted too
        key = self.optim_key if self.optim_key in ckpt else ("w_optim" if "w_optim" in ckpt else "g_optim")
        self.optimizer.load_state_dict(ckpt[key])
        return start_iter

    def save(self, idx: int, checkpoint_path: str) -> str:
        path = f"{checkpoint_path}/{str(idx).zfill(6)}.pt"
        torch.save({"gen": self.gen.state_dict(), self.optim_key: self.optimizer.state_dict(), "args": self.args}, path)
        return path


def fit_frames(trainer: "Trainer", reals: torch.Tensor, labels: torch.Tensor, params: Optional[torch.Tensor] = None,
               epochs: int = 1, batch: Optional[int] = None, tune_iter: Optional[int] = None,
               start_iter: int = 0, on_step: Optional[Callable] = None,
               uniforms: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> List[torch.Tensor]:
    """The fitting loop of train_rgb.py:114-154 / train_3dmm.py:113-150 over a frame set resident on the device:
    frames are sharded in contiguous blocks over the ranks (`shard_range`), every rank walks its shard in
    batches of `batch` (default: args.batch_size // world_size, at least 1 — the reference's integer division can
    give 0), ragged tails are handled by `epoch_batches`, the shared gradients are averaged by the
    trainer's one flat all-reduce, and the generator starts being tuned once `i + 1 >= tune_iter`.
    `reals` [N,3,s,s], `labels` [N,25] (un-flipped, as the data set yields them), `params` [N,P] (3dmm mode).
    `uniforms` (TEST HOOK) = (u_strat [N,R,Sc], u_imp [N,R,Sf]): per-FRAME renderer uniforms, so that a frame samples alike
    whichever rank and batch position renders it (exact multi-rank equalities); default: fresh draws per call.
    Returns the per-step L2 losses of THIS rank (device scalars; no host sync inside the loop)."""
    tr = trainer
    n = reals.shape[0]
    if batch is None:
        batch = max(1, tr.batch_size // max(tr.world_size, 1))
    losses: List[torch.Tensor] = []
    i = start_iter
    for _ in range(epochs):
        for idx, weight in epoch_batches(n, tr.rank, tr.world_size, batch):
            # a rank's batch is a contiguous block of frames: slice views, no index tensor (uploading one per step from
            # pageable memory blocks the host until the launch stream has drained — 1.8 ms per step of lost run-ahead)
            lo = int(idx[0]) if idx.numel() else 0
            sl = slice(lo, lo + idx.numel())
            real, label = reals[sl], labels[sl].clone()         # the label flip is in place: never on the data set
            kw = {}
            if uniforms is not None and idx.numel():
                kw = dict(u_strat=uniforms[0][sl].contiguous(),
                          u_imp=uniforms[1][sl].reshape(-1, uniforms[1].shape[-1]).contiguous())
            if tr.mode == "rgb":
                out = tr.gen_update(real, label, loss_weight=weight, **kw)
                l2 = out[0]
            else:
                out = tr.gen_update(real, label, params[sl], loss_weight=weight, **kw)
                l2 = out[1]
            losses.append(l2)
            if on_step is not None:
                on_step(i, out)
            if tune_iter is not None and (i + 1) >= tune_iter:
                tr.tune_generator()
            i += 1
    return losses


def audio_window(auds: torch.Tensor, img_i: int, smo_size: int, limit: int) -> torch.Tensor:
    """The smoothing window of frame ``img_i``: auds[img_i - h : img_i + h] (h = smo_size // 2) clipped to
    [0, limit) and zero-padded back to smo_size rows — trainer_audio.py:66-83 (limit = i_train) / :127-144
    (limit = len(auds))."""
    half = int(smo_size / 2)
    left, right = img_i - half, img_i + half
    pad_left = pad_right = 0
    if left < 0:
        pad_left, left = -left, 0
    if right > limit:
        pad_right, right = right - limit, limit
    win = auds[left:right]
    if pad_left > 0:
        win = torch.cat((torch.zeros_like(win)[:pad_left], win), dim=0)
    if pad_right > 0:
        win = torch.cat((win, torch.zeros_like(win)[:pad_right]), dim=0)
    return win


def audio_windows(auds: torch.Tensor, idx: torch.Tensor, smo_size: int, limit: int) -> torch.Tensor:
    """Batched `audio_window`: idx [N] → [N, smo_size, ...]; one gather instead of N slices.  Reproduces the
    reference's padding exactly, including its quirk that a pad longer than the clipped window is truncated
    (``zeros_like(win)[:pad]``) — callers keep ``half <= limit``."""
    half = int(smo_size / 2)
    off = torch.arange(-half, half, device=idx.device)
    pos = idx[:, None] + off[None, :]                                  # [N, smo_size]
    valid = (pos >= 0) & (pos < limit)
    win = auds[pos.clamp(0, limit - 1)]
    return win * valid.view(*valid.shape, *([1] * (win.dim() - 2))).to(win.dtype)


class AudioTrainer(nn.Module):
    """Counterpart of /root/reference/code/trainer_audio.py:20-217: HeadNeRF_Audio driven by AudioNet (+ AudioAttNet
    over a `smo_size` window once `global_step >= nosmo_iters`), three Adam optimisers, the reference's checkpoint
    keys.  Multi-GPU as in `Trainer`: one flattened all-reduce of the shared gradients (latent basis, 3DMM net is
    unused here, AudioNet, AudioAttNet; generator once tuned) per step instead of the reference's three DDP wrappers."""

    def __init__(self, auds, i_train: int, args, device, rank: int = 0, world_size: int = 1,
                 lpips: Optional[Callable] = None, gen: Optional[nn.Module] = None):
        super().__init__()
        self.args, self.device, self.rank, self.world_size = args, device, rank, world_size
        self.batch_size = args.batch_size
        if gen is None:
            gen = HeadNeRF_Audio(args, args.size, device, args.latent_dim_style, args.latent_dim_shape,
                                 getattr(args, "run_id", "nerface2"), getattr(args, "emb_dir", "./PTI/embeddings/"))
        self.gen = gen.to(device)
        self.AudNet = AudioNet(args.dim_aud, args.win_size).to(device)
        self.AudAttNet = AudioAttNet().to(device)       # default dim_aud = 32 while features are 64-d: reference quirk 7
        self.optimizer_Aud = _adam(list(self.AudNet.parameters()), lr=args.lr, betas=(0.9, 0.999))
        self.optimizer_AudAtt = torch.optim.Adam(params=list(self.AudAttNet.parameters()), lr=args.lr,
                                                 betas=(0.9, 0.999))
        self.w_optim = _adam(self.gen.parameters(), lr=args.lr)
        requires_grad(self.gen.generator, False)
        if lpips is None:
            warnings.warn("AudioTrainer: no LPIPS module given — optimising the L2 term only (the reference trains on "
                          "l2 + LPIPS(alex)); pass lpips=LPIPSAlex(state_dict) or lpips='none' to silence this",
                          stacklevel=2)
        self.lpips_loss = None if (lpips is None or (isinstance(lpips, str) and lpips == "none")) else lpips
        self._flat: Optional[FlatGrads] = None
        self.auds = torch.as_tensor(auds).to(device).float()
        self.i_train = i_train
        self.face_pool = nn.AdaptiveAvgPool2d((args.size, args.size))
        if world_size > 1:
            import torch.distributed as dist
            with torch.no_grad():
                for m in (self.gen, self.AudNet, self.AudAttNet):
                    for t in list(m.parameters()) + list(m.buffers()):
                        dist.broadcast(t, src=0)
            inv = getattr(self.gen.generator, "invalidate_caches", None)
            if inv is not None:
                inv()

    def shared_parameters(self):
        return _readiness_order(self.gen, self.AudNet, self.AudAttNet)

    def flat_grads(self) -> FlatGrads:
        shared = self.shared_parameters()
        if self._flat is None or not self._flat.owns(shared):
            self._flat = FlatGrads(shared, getattr(self, "bucket_bytes", 32 << 20))
            self._bucketer = None
        return self._flat

    def _overlap(self, flat: FlatGrads) -> Optional[BucketedAllReduce]:
        if self.world_size <= 1 or flat.flat is None:
            return None
        if getattr(self, "_bucketer", None) is None:
            self._bucketer = BucketedAllReduce(flat, self.world_size)
            _install_grad_hooks(self, [self.gen, self.AudNet, self.AudAttNet])
        return self._bucketer

    def l2_loss(self, real_images, generated_images):
        return F.mse_loss(real_images, generated_images, reduction="mean")

    def tune_generator(self):
        requires_grad(self.gen.generator, True)

    def _drive(self, global_step: int, img_i: int, limit: int) -> torch.Tensor:
        """Audio feature of frame img_i → driver input of HeadNeRF_Audio [1, dim_aud] (trainer_audio.py:65-94)."""
        # img_i: the data loader's 1-element index tensor in the reference; a plain int is accepted as well
        if global_step >= self.args.nosmo_iters:
            feats = self.AudNet(audio_window(self.auds, int(img_i), self.args.smo_size, limit))
            aud = self.AudAttNet(feats)
        else:
            aud = self.AudNet(self.auds[img_i].reshape(-1, *self.auds.shape[1:]).squeeze(1))
        return aud.unsqueeze(0) if aud.dim() == 1 else aud

    def gen_update(self, real_image, label, params, global_step: int, img_i: int, person_2: bool = False, *,
                   u_strat: Optional[torch.Tensor] = None, u_imp: Optional[torch.Tensor] = None):
        """trainer_audio.py:55-113.  `u_strat` / `u_imp`: the renderer-uniform test hook of `Trainer.gen_update`."""
        self.gen.train(), self.AudNet.train(), self.AudAttNet.train()
        flat = self.flat_grads()
        bucketer = self._overlap(flat)
        if bucketer is not None:
            bucketer.reset()
        flat.zero()
        smooth = global_step >= self.args.nosmo_iters
        absent = _generator_idle_parameters(self.gen.generator)
        if not smooth:                     # the attention net is not on the path yet (trainer_audio.py:88-94)
            absent = absent + [p for p in self.AudAttNet.parameters() if p.requires_grad]
        with _StepScope(bucketer, [self.gen.generator], absent):
            generated = self.gen(self._drive(global_step, img_i, self.i_train), label, person_2,
                                 u_strat=u_strat, u_imp=u_imp)
            l2_3dmm = torch.zeros(1, device=self.device)
            l2, generated = pooled_l2(self.face_pool, real_image, generated, self.lpips_loss is not None)
            lp = (torch.squeeze(self.lpips_loss(real_image, generated)).mean() if self.lpips_loss is not None
                  else torch.zeros((), device=l2.device))
            (l2_3dmm + l2 + lp).backward()
            if bucketer is not None:
                bucketer.finish()
        opts = [self.w_optim, self.optimizer_Aud] + ([self.optimizer_AudAtt] if smooth else [])
        step_skipping(opts, absent)
        return l2_3dmm, l2.detach(), lp.detach(), generated.detach()

    def sample(self, real_image, label, params, global_step: int, img_i: int, person_2: bool = False):
        with torch.no_grad():
            self.gen.eval(), self.AudNet.eval(), self.AudAttNet.eval()
            return self.gen(self._drive(global_step, img_i, self.auds.shape[0]), label, person_2)

    @torch.no_grad()
    def sample_frames(self, idx: torch.Tensor, label: torch.Tensor, person_2: bool = False) -> torch.Tensor:
        """Batched `sample` for the reenactment harness (BASELINE config 5): frames idx [N] with the smoothing
        window, ONE AudioNet pass over the N*smo_size windows, one batched attention, one synthesis of N frames."""
        self.gen.eval(), self.AudNet.eval(), self.AudAttNet.eval()
        return self.gen(self.drive_frames(idx), label, person_2)

    def drive_frames(self, idx: torch.Tensor) -> torch.Tensor:
        """Smoothed audio features of frames idx [N] → [N, dim_aud]; row n equals `_drive` of frame idx[n]."""
        n, smo = idx.shape[0], self.args.smo_size
        win = audio_windows(self.auds, idx, smo, self.auds.shape[0])              # [N, smo, 16, 29]
        feats = self.AudNet(win.reshape(n * smo, *win.shape[2:])).reshape(n, smo, -1)
        return self.AudAttNet.forward_windows(feats)

    def sample_bases(self, person_2: bool = False):
        """trainer_audio.py:154-176: alpha = 5 e_i, frontal camera, ONE label tensor re-used (alternating flip)."""
        imgs = []
        with torch.no_grad():
            pts, _, _ = sample_camera_positions(device=self.device, n=1, r=2.7, horizontal_mean=0.5 * math.pi,
                                                vertical_mean=0.5 * math.pi, mode=None)
            label = make_label(create_cam2world_matrix(-pts, pts, device=self.device))
            self.gen.eval()
            k = self.args.latent_dim_shape
            for i in range(k):
                w = torch.zeros(1, k, device=self.device)
                w[0, i] = 5
                imgs.append(self.gen.get_image(self.gen.get_latent(w, person_2), label))
        return imgs

    def resume(self, resume_ckpt: str) -> int:
        ckpt = torch.load(resume_ckpt, map_location=self.device, weights_only=False)
        start_iter = int(os.path.splitext(os.path.basename(resume_ckpt))[0])
        self.gen.load_state_dict(ckpt["gen"])
        self.AudNet.load_state_dict(ckpt["AudNet"])
        self.AudAttNet.load_state_dict(ckpt["AudAttNet"])
        self.w_optim.load_state_dict(ckpt["w_optim"])
        self.optimizer_Aud.load_state_dict(ckpt["optimizer_Aud"])
        self.optimizer_AudAtt.load_state_dict(ckpt["optimizer_AudAtt"])
        return start_iter

    def save(self, idx: int, checkpoint_path: str) -> str:
        path = f"{checkpoint_path}/{str(idx).zfill(6)}.pt"
        torch.save({"gen": self.gen.state_dict(), "AudAttNet": self.AudAttNet.state_dict(),
                    "AudNet": self.AudNet.state_dict(), "w_optim": self.w_optim.state_dict(),
                    "optimizer_Aud": self.optimizer_Aud.state_dict(),
                    "optimizer_AudAtt": self.optimizer_AudAtt.state_dict(), "args": self.args}, path)
        return path

"""TriPlaneGenerator: the object HFA-GP holds as ``self.generator``.

Drop-in surface: ``synthesis(ws, c, noise_mode='const') ->
{'image','image_raw','image_depth'}``, ``parameters()``, ``requires_grad_``,
``to()``, ``state_dict()`` with EG3D key names (``backbone.synthesis.b8.conv0.weight``
...), exactly what ``load_G_official`` (/root/reference/code/networks/headnerf.py:31-38)
hands to ``HeadNeRF_*`` and what ``trainer_rgb.py:59-60,70-71,143-151`` touch.

The arithmetic runs in libhfagp_hip.so (see ops.py).  There is no PyTorch
fallback: calling ``synthesis`` with CPU tensors or without the library raises.

The renderer's random draws (EG3D uses ``torch.rand_like`` / ``torch.rand``
inside ``ImportanceRenderer`` even in eval mode) are drawn here from the torch
device generator, or passed in as ``u_strat`` / ``u_imp`` for reproducible parity.
"""
from __future__ import annotations

import contextlib

import math
import os
from typing import Dict, Optional

import torch
from torch import nn

from . import ops
from .config import GeneratorConfig, ffhq512_128


# --------------------------------------------------------------------------- parameter containers
class _Affine(nn.Module):
    def __init__(self, w_dim: int, cin: int, gen: torch.Generator):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(cin, w_dim, generator=gen))
        self.bias = nn.Parameter(torch.ones(cin))           # bias_init = 1


class _SynthesisLayer(nn.Module):
    def __init__(self, cin, cout, w_dim, resolution, up, taps, gen, use_noise=True):
        super().__init__()
        self.up, self.resolution = up, resolution
        self.affine = _Affine(w_dim, cin, gen)
        self.weight = nn.Parameter(torch.randn(cout, cin, 3, 3, generator=gen))
        if use_noise:
            self.register_buffer("noise_const", torch.randn(resolution, resolution, generator=gen))
            self.noise_strength = nn.Parameter(torch.zeros([]))
        self.bias = nn.Parameter(torch.zeros(cout))
        self.register_buffer("resample_filter", _fir(taps))


class _ToRGB(nn.Module):
    def __init__(self, cin, cout, w_dim, gen):
        super().__init__()
        self.affine = _Affine(w_dim, cin, gen)
        self.weight = nn.Parameter(torch.randn(cout, cin, 1, 1, generator=gen))
        self.bias = nn.Parameter(torch.zeros(cout))


class _SynthesisBlock(nn.Module):
    def __init__(self, cin, cout, w_dim, resolution, img_channels, taps, gen):
        super().__init__()
        self.in_channels, self.out_channels, self.resolution = cin, cout, resolution
        if cin == 0:
            self.const = nn.Parameter(torch.randn(cout, resolution, resolution, generator=gen))
        else:
            self.conv0 = _SynthesisLayer(cin, cout, w_dim, resolution, 2, taps, gen)
        self.conv1 = _SynthesisLayer(cout, cout, w_dim, resolution, 1, taps, gen)
        self.torgb = _ToRGB(cout, img_channels, w_dim, gen)
        self.register_buffer("resample_filter", _fir(taps))


class _FC(nn.Module):
    def __init__(self, fin, fout, gen, lr_mul=1.0, bias_init=0.0):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(fout, fin, generator=gen) / lr_mul)
        self.bias = nn.Parameter(torch.full([fout], float(bias_init)))


def _fir(taps) -> torch.Tensor:
    k = torch.tensor(list(taps), dtype=torch.float32)
    k = torch.outer(k, k)
    return k / k.sum()


class _Synthesis(nn.Module):
    pass


class _Backbone(nn.Module):
    pass


class _Mapping(nn.Module):
    pass


class _SuperRes(nn.Module):
    pass


class _Decoder(nn.Module):
    pass


# --------------------------------------------------------------------------- the generator
class TriPlaneGenerator(nn.Module):
    def __init__(self, cfg: Optional[GeneratorConfig] = None, seed: int = 0):
        super().__init__()
        self.cfg = cfg = cfg or ffhq512_128()
        cfg.validate()
        if tuple(cfg.resample_filter) != (1, 3, 3, 1):
            raise ValueError("the fused up-sampling kernels are written for resample_filter [1,3,3,1]")
        gen = torch.Generator().manual_seed(seed)
        taps = cfg.resample_filter
        # EG3D attribute names (for state_dict compatibility)
        self.z_dim, self.c_dim, self.w_dim = cfg.z_dim, cfg.c_dim, cfg.w_dim
        self.img_resolution, self.img_channels = cfg.img_resolution, cfg.img_channels
        self.neural_rendering_resolution = cfg.neural_rendering_resolution

        self.backbone = _Backbone()
        syn = _Synthesis()
        for res in cfg.block_resolutions:
            cin = cfg.channels(res // 2) if res > 4 else 0
            setattr(syn, f"b{res}", _SynthesisBlock(cin, cfg.channels(res), cfg.w_dim, res,
                                                    cfg.backbone_img_channels, taps, gen))
        self.backbone.synthesis = syn
        mp = _Mapping()
        mp.embed = _FC(cfg.c_dim, cfg.w_dim, gen)
        for i in range(cfg.mapping_layers):
            setattr(mp, f"fc{i}", _FC(2 * cfg.w_dim if i == 0 else cfg.w_dim, cfg.w_dim, gen, cfg.mapping_lr_mul))
        mp.register_buffer("w_avg", torch.zeros(cfg.w_dim))
        self.backbone.mapping = mp

        sr = _SuperRes()
        r0, r1 = cfg.sr_resolutions
        sr.block0 = _SynthesisBlock(cfg.plane_channels, cfg.sr_channels[0], cfg.w_dim, r0, cfg.img_channels, taps, gen)
        sr.block1 = _SynthesisBlock(cfg.sr_channels[0], cfg.sr_channels[1], cfg.w_dim, r1, cfg.img_channels, taps, gen)
        self.superresolution = sr

        dec = _Decoder()
        dec.net = nn.ModuleDict({
            "0": _FC(cfg.plane_channels, cfg.decoder_hidden, gen, cfg.decoder_lr_mul),
            "2": _FC(cfg.decoder_hidden, 1 + cfg.plane_channels, gen, cfg.decoder_lr_mul),
        })
        self.decoder = dec
        self._prep: Dict[object, tuple] = {}     # (kind, ..., id(param)) -> (version, data_ptr, image, wsq)
        self._conv_precision = "fp32"
        self._sr_conv_precision: Optional[str] = None
        self.conv_precision = cfg.conv_precision
        self.sr_conv_precision = cfg.sr_conv_precision
        self.sr_storage = cfg.sr_storage
        # up-sampling layers in ONE pass (FIR fused into the transposed conv, csrc/upconv_fir.hip).  Measured on the MI355X
        # (profiles/r03_upfir/): the FIR arithmetic runs in the same waves as the GEMM, so the fused form only wins where the
        # GEMM is short and the layer is dominated by the round trip of the raw result through HBM — Cin <= 64, i.e. the first
        # super-resolution layer (32 -> 256 @128^2: 1.43 ms against 0.60 + 1.00 ms at B = 32); it ties or loses by 2 - 5 % on the
        # 256- and 512-channel layers.  "auto" (default): fused for Cin <= 64 where the launch fills the chip; "1": wherever the
        # library supports it; "0": never.
        self.fuse_up_fir = os.environ.get("HFAGP_FUSE_UP_FIR", "auto")
        # Image side chain (toRGB + skip of block k, and its adjoints) on a SECOND HIP stream for batches up to this value: same bits,
        # +1 % on one forward frame, two pathologies in fitting loops (allocator churn from record_stream marks; pool-stream aliasing
        # with collectives) — OFF by default (0); DESIGN.md section 8, round 4.
        self.side_stream_max_batch = int(os.environ.get("HFAGP_SIDE_STREAM_MAX_BATCH", "0"))
        self._side_streams: Dict[int, "torch.cuda.Stream"] = {}
        self._styles: Dict[int, tuple] = {}      # id(layer) -> (styles, dcoef) of the pass in flight
        self._absmax = None                      # (slot buffers, layer names) of the last pass: f16_range_report()
        self._rgb_part = None                    # partial toRGB sums of the conv just run (fused toRGB, ops.modconv)
        self._planes_absmax = None               # [64] slots with max |planes| of the pass in flight (16-bit decoder)
        self._scalars: Dict[int, tuple] = {}     # id(param) -> (version, data_ptr, python float)
        self._const_nhwc: Optional[tuple] = None
        self.timing: Optional[Dict[str, list]] = None   # bench.py: {'raymarch': [(ev0, ev1, units)], 'modconv': [...]}
        self.ignored_checkpoint_keys: list = []  # EG3D entries a loaded checkpoint carried that have no tensor here
        self._register_load_state_dict_pre_hook(self._drop_foreign_eg3d_keys)
        # a loaded checkpoint replaces parameters AND buffers in place: every derived image goes (the flat noise image)
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: module.invalidate_caches())

    # entries of an EG3D `G_ema.state_dict()` / HFA-GP `ckpt["gen"]["generator.*"]` that carry no parameter of the
    # path: version-dependent helper buffers of modules that are fused away here.  They are dropped (and listed in
    # `ignored_checkpoint_keys`) so that the reference's STRICT `load_state_dict` calls (trainer_rgb.py:137) succeed;
    # any other unexpected or missing key still raises.
    FOREIGN_KEY_PREFIXES = ("renderer.", "ray_sampler.", "superresolution.resample_filter")

    def _drop_foreign_eg3d_keys(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                error_msgs):
        drop = [k for k in state_dict if k.startswith(prefix) and k[len(prefix):].startswith(self.FOREIGN_KEY_PREFIXES)]
        for k in drop:
            del state_dict[k]
        self.ignored_checkpoint_keys = [k[len(prefix):] for k in drop]

    def side_stream(self, batch: int, device) -> Optional["torch.cuda.Stream"]:
        """The second launch stream of `device` for the image side chain, or None when the batch is too large for it to pay
        (or while a HIP graph is being captured on the launch stream: a capture owns one stream)."""
        if batch > self.side_stream_max_batch or torch.cuda.is_current_stream_capturing():
            return None
        idx = device.index if device.index is not None else torch.cuda.current_device()
        st = self._side_streams.get(idx)
        if st is None:
            # from the HIGH-priority pool: PyTorch hands out its 32 default-priority pool streams round robin, and the collective
            # back ends take theirs from that same pool (gloo: one per asynchronous work) — a side stream from it ends up ALIASED
            # with a collective's copy stream every few steps and the image chain then queues behind host-side all-reduces
            # (measured: the 2-rank RGB fitting step 49 ms -> 2982 ms).  The high-priority pool is not used by them.
            st = self._side_streams[idx] = torch.cuda.Stream(device=device, priority=-1)
        return st

    def _timed(self, key: str, units: float, fn, *args, **kwargs):
        """Run ``fn`` bracketed by HIP events on the current stream when bench.py enabled timing."""
        if self.timing is None:
            return fn(*args, **kwargs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*args, **kwargs)
        e1.record()
        self.timing.setdefault(key, []).append((e0, e1, units))
        return out

    # ----------------------------------------------------------------- caches
    def invalidate_caches(self) -> None:
        """Drop every derived image of the parameters (GEMM weight images, wsq, host copies of scalars, the NHWC
        constant, the flat noise image).  The caches are keyed by (id, `_version`, `data_ptr`) of the parameter, which catches
        torch's own in-place ops, `load_state_dict` and `.to()` — and `MultiTensorAdam`, which bumps the versions itself; it does
        NOT catch `torch.optim.Adam(fused=True)` (leaves `_version` alone: the round-5 stale-image find) nor an in-place write through
        `.data` (EMA / PTI-style `.data.copy_`).  Hence: while any parameter requires grad every forward rebuilds the images
        (`_refresh_tuned`), the tuned -> frozen transition of the same object drops everything once, and callers that write through
        `.data` while frozen must call this.  Also called by `_apply` (device / dtype moves), `__deepcopy__` and after
        `load_state_dict` (post hook).  DESIGN.md section 5.2 lists every cache and its regression test."""
        self._prep = {}
        self._scalars = {}
        self._const_nhwc = None
        self._styles = {}
        self._plist = None
        self._conv_weights = None
        self._noise_layers = None
        self._noise_key = None
        self._scaled_noise = {}

    def _refresh_tuned(self, backward_follows: bool = True) -> None:
        """Start of every forward pass while ANY parameter requires grad (the generator is being tuned, trainer_rgb.py:69-71): drop the
        weight images / wsq / NHWC constant and re-read the 0-d parameters.  The caches are keyed by the parameters' `_version`, and an
        optimiser's in-place update does not reliably advance it: `torch.optim.Adam(fused=True)` — what the trainers build — moves
        every weight by lr per step and leaves `_version` at 0 (measured, round 5: rounds 2-4 ran the tuned step on the weight
        images of step 0, and timed it without the per-step image rebuild).  While the generator is frozen nothing here runs.
        The host copies of the noise strengths are refreshed in ONE device-to-host copy per step."""
        plist = getattr(self, "_plist", None)
        if plist is None:
            plist = self._plist = [p for n, p in self.named_parameters() if not n.startswith("backbone.mapping.")]
            self._scalar_params = [p for p in plist if p.dim() == 0]
        if not any(p.requires_grad for p in plist):
            self._scaled_noise = {}
            if getattr(self, "_was_tuned", False):
                # tuned -> frozen on the SAME object: what the last tuned forward cached predates the last optimiser step,
                # and the host copies of the noise strengths were never refreshed while tuned
                self._was_tuned = False
                self.invalidate_caches()
            return
        self._was_tuned = True
        self._prep = {}
        self._const_nhwc = None
        # noise strengths: NO host copy while tuned (a device-to-host read per step stalls the launch queue: +1.1 ms per step measured).
        # The forward kernels take noise_const * strength as their noise image with strength 1 (one gather + multiply over the 830 k
        # noise values of all layers), the fused backward pass reads the strength from device memory (noise_strength_dev).
        nl = getattr(self, "_noise_layers", None)
        # (the flat copy of the noise buffers follows them: load_state_dict / Trainer.resume copy into the buffers in place)
        nkey = None if nl is None else tuple((m.noise_const._version, m.noise_const.data_ptr()) for m in nl)
        if nl is None or nkey != getattr(self, "_noise_key", None):
            nl = self._noise_layers = [m for m in self.modules() if isinstance(m, _SynthesisLayer) and getattr(m, "noise_const", None) is not None]
            self._noise_key = tuple((m.noise_const._version, m.noise_const.data_ptr()) for m in nl)
            if nl:
                self._noise_flat = torch.cat([m.noise_const.detach().reshape(-1) for m in nl])
                self._noise_idx = torch.cat([torch.full((m.noise_const.numel(),), i, dtype=torch.long, device=self._noise_flat.device)
                                             for i, m in enumerate(nl)])
        self._scaled_noise = {}
        if nl:
            strengths = torch.stack([m.noise_strength.detach() for m in nl])
            scaled = self._noise_flat * strengths[self._noise_idx]
            off = 0
            for m in nl:
                n = m.noise_const.numel()
                self._scaled_noise[id(m)] = scaled[off: off + n].view_as(m.noise_const)
                off += n
        # ... and everything the step needs of the conv weights in ONE launch (ops.weight_prep_batch): forward image, image of the
        # transpose (when a backward pass can follow) and wsq; per-layer launches were 1.2 ms of a 15 ms step
        convs = getattr(self, "_conv_weights", None)
        if convs is None:
            convs = self._conv_weights = [(p, n.endswith((".conv0.weight", ".conv1.weight"))) for n, p in self.named_parameters()
                                          if n.endswith((".conv0.weight", ".conv1.weight", ".torgb.weight")) and p.dim() == 4]
        want_t = backward_follows        # (a call that records a tape: the bwd-data GEMMs will ask for the transposed images)
        items, owners = [], []
        for w, is_layer in convs:
            if not (w.is_cuda and ops.weight_prep_batch_supported(w)):
                continue
            prec, prec_t = self._precision_of(w), self._precision_of(w, True)
            prec = "f16x3" if prec == "f16x2" else prec
            prec_t = "f16x3" if prec_t == "f16x2" else prec_t
            if prec == "fp32":
                continue
            items.append((w.detach(), prec, prec_t if (want_t and prec_t != "fp32") else None, is_layer))
            owners.append((w, prec, prec_t))
        if items:
            for (w, prec, prec_t), (img, img_t, wsq) in zip(owners, ops.weight_prep_batch(items)):
                self._prep[("G", prec, False, id(w))] = (w._version, w.data_ptr(), img, None)
                if img_t is not None:
                    self._prep[("G", prec_t, True, id(w))] = (w._version, w.data_ptr(), img_t, None)
                if wsq is not None:
                    self._prep[("Q", id(w))] = (w._version, w.data_ptr(), None, wsq)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if hasattr(self, "_prep"):
            self.invalidate_caches()
        return out

    def __deepcopy__(self, memo):
        """`copy.deepcopy(generator)` (the reference's load_G_official deep-copies G_ema, headnerf.py:35): parameters,
        buffers and the precision settings are copied, the derived caches are not (their keys are object ids)."""
        import copy
        saved = (self._prep, self._scalars, self._const_nhwc, self._styles, self.timing)
        self.invalidate_caches()
        self.timing = None
        try:
            new = self.__class__.__new__(self.__class__)
            memo[id(self)] = new
            new.__dict__ = copy.deepcopy(self.__dict__, memo)
        finally:
            self._prep, self._scalars, self._const_nhwc, self._styles, self.timing = saved
        return new

    def _is_sr_weight(self, weight: torch.Tensor) -> bool:
        """True for the 3x3 conv weights of the two super-resolution blocks (resolved by module, not by a cached id)."""
        sr = self.superresolution
        return any(weight is l.weight for blk in (sr.block0, sr.block1) for l in (blk.conv0, blk.conv1))

    @property
    def conv_precision(self) -> str:
        """Arithmetic of the conv GEMMs: 'fp32' (exact MFMA), 'bf16x3' / 'bf16x6' / 'f16x3' (split-operand MFMA, fp32
        accumulation; include/hfagp.h HFAGP_PREC_*), 'f16' (single-pass fp16 MFMA, fp32 accumulation: the
        arithmetic of EG3D's fp16 blocks).  'f16x3' (the default) splits each operand into two fp16 parts (22 mantissa
        bits): fp32-class results at the cost of 'bf16x3'.  'f16x2' (opt-in): the 22-bit weights against activations rounded
        to ONE fp16 part, two MFMAs per product — the class of TF32, which the reference's cuDNN convs run in on Ampere-class
        GPUs (toRGB layers and gradient GEMMs stay as under 'f16x3').  Layers whose shape the 16-bit kernels do not take
        (Cin % 16, Cout % 128) always run on the exact fp32 kernel; gradient GEMMs of the fp16 kinds run in bf16x3 (a
        raw gradient has no place in fp16's exponent range without loss scaling)."""
        return self._conv_precision

    @property
    def sr_conv_precision(self) -> Optional[str]:
        """Precision of the super-resolution blocks' 3x3 convs; None = conv_precision."""
        return self._sr_conv_precision

    @sr_conv_precision.setter
    def sr_conv_precision(self, value: Optional[str]):
        if value is not None and value not in ops.PRECISIONS:
            raise ValueError(f"sr_conv_precision must be None or one of {sorted(ops.PRECISIONS)}, got {value!r}")
        self._sr_conv_precision = value

    @conv_precision.setter
    def conv_precision(self, value: str):
        if value not in ops.PRECISIONS:
            raise ValueError(f"conv_precision must be one of {sorted(ops.PRECISIONS)}, got {value!r}")
        self._conv_precision = value

    def _precision_of(self, weight: torch.Tensor, transposed: bool = False) -> str:
        """The precision a conv weight's GEMM runs in (the Cin/Cout transpose: the bwd-data GEMM)."""
        co, ci = weight.shape[:2]
        if transposed:
            co, ci = ci, co
        prec = self._conv_precision
        if self._sr_conv_precision is not None and self._is_sr_weight(weight):
            prec = self._sr_conv_precision
        if prec in ("f16", "f16x3", "f16x2") and transposed:
            prec = "bf16x3"       # gradient GEMMs: a raw gradient needs fp32's exponent range (bf16 parts have it)
        if prec in ("f16", "f16x2") and weight.shape[-1] == 1:
            prec = "f16x3"        # the toRGB products feed the tri-planes directly: keep them fp32-class
        if not ops.split_supported(ci, co):
            prec = "fp32"
        return prec

    def _gemm_image(self, weight: torch.Tensor, transposed: bool = False) -> torch.Tensor:
        """MFMA B-operand image of a conv weight (of its Cin/Cout transpose for the bwd-data GEMMs), in the layout
        of the configured precision; cached until the parameter changes."""
        prec = self._precision_of(weight, transposed)
        if prec == "f16x2":
            prec = "f16x3"        # same image: two fp16 parts of the weights (the ACTIVATIONS are the single part)
        key = ("G", prec, transposed, id(weight))
        hit = self._prep.get(key)
        if hit is not None and hit[0] == weight._version and hit[1] == weight.data_ptr():
            return hit[2]
        w = weight.detach()
        w = (w.transpose(0, 1) if transposed else w).contiguous()
        img = ops.weight_prep_prec(w, prec) if prec != "fp32" else ops.weight_prep(w)[0]
        self._prep[key] = (weight._version, weight.data_ptr(), img, None)
        return img

    def _prepared(self, weight: torch.Tensor):
        """(GEMM image, wsq [Cout, Cin] = sum over taps of weight^2 for the demodulation)."""
        key = ("Q", id(weight))
        hit = self._prep.get(key)
        if hit is None or hit[0] != weight._version or hit[1] != weight.data_ptr():
            _, wsq = ops.weight_prep(weight.detach().contiguous())
            hit = (weight._version, weight.data_ptr(), None, wsq)
            self._prep[key] = hit
        return self._gemm_image(weight), hit[3]

    def _scalar(self, t: torch.Tensor) -> float:
        """Host copy of a 0-d parameter (noise_strength), cached so steady state has no device sync."""
        hit = self._scalars.get(id(t))
        if hit is not None and hit[0] == t._version and hit[1] == t.data_ptr():
            return hit[2]
        v = float(t.detach())
        self._scalars[id(t)] = (t._version, t.data_ptr(), v)
        return v

    def _const(self, const: torch.Tensor) -> torch.Tensor:
        hit = self._const_nhwc
        if hit is not None and hit[0] == const._version and hit[1] == const.data_ptr():
            return hit[2]
        x = ops.nchw_to_nhwc(const.detach()[None].contiguous())
        self._const_nhwc = (const._version, const.data_ptr(), x)
        return x

    # ----------------------------------------------------------------- layers
    def _layer(self, x, layer: _SynthesisLayer, w, row, batch, noise_mode, conv_clamp, tape, x_absmax=None,
               y_absmax=None, rgb=None, half=False, keep_out=True):
        """x_absmax / y_absmax: fp16 range tracking of an UNCLAMPED activation chain (ops.modconv): the slot buffer with
        max |x| of the input as published by its producer, and the one this layer publishes max |out| into.
        half: write the output (and the raw up-conv intermediate) as float16 (`sr_storage = "f16"`, no tape)."""
        cfg = self.cfg
        wt, wsq = self._prepared(layer.weight)
        pre = self._styles.pop(id(layer), None) if self._styles else None      # computed up front (_precompute_styles)
        styles, dcoef = pre if pre is not None else ops.styles_demod(w, layer.affine.weight, layer.affine.bias, wsq,
                                                                     1.0, cfg.demod_eps)
        noise, ns = None, 0.0
        noise_raw, ns_dev = None, None
        if noise_mode == "const":
            scaled = getattr(self, "_scaled_noise", {}).get(id(layer))
            if scaled is not None:       # generator being tuned (_refresh_tuned): pre-scaled noise image, strength stays on the device
                noise, ns, noise_raw, ns_dev = scaled, 1.0, layer.noise_const, layer.noise_strength.detach()
            else:
                noise, ns = layer.noise_const, self._scalar(layer.noise_strength)
                noise_raw = noise
        elif noise_mode != "none":
            raise NotImplementedError("noise_mode must be 'const' or 'none' (HFA-GP passes 'const', headnerf.py:112)")
        cout = layer.weight.shape[0]
        gain = math.sqrt(2.0)
        x_parts = 1 if self._precision_of(layer.weight) == "f16x2" else 0
        k_styles, k_dcoef = styles, dcoef       # (the fp16 kernels apply EG3D's fp16 range guard themselves)
        # algorithmic FLOPs: 2 * B * H_in * W_in * Cin * Cout * 9 (the up-conv is counted in its
        # polyphase / transposed form at INPUT resolution)
        flops = 2.0 * batch * x.shape[1] * x.shape[2] * x.shape[3] * cout * 9
        # which kernel bench.py times
        key = ("modconv" if wt.dtype == torch.float32 else
               "modconv_f16" if wt.dtype == torch.float16 and wt.shape[0] == 1 else "modconv_split")
        if (layer.up != 2 and wt.dtype != torch.float32 and x.shape[1] * x.shape[2] <= 256 and x.shape[3] % 16 == 0
                and x.shape[3] <= 512 and cout % 32 == 0 and not half and rgb is None):
            # the library runs these on smallconv_kernel (csrc/smallconv.hip, modconv_plan.h `smallconv_takes` — same predicate:
            # fp16-storage, Cin > 512 and fused-toRGB layers stay on the staged kernel): not the roofline kernel
            key = "modconv_small"
        fuse = self.fuse_up_fir
        # (fp16 STORAGE halves the stand-alone FIR kernel's traffic: there the two-kernel form measured 0.6 % ahead)
        fuse = fuse in (True, "1") or (fuse == "auto" and x.shape[3] <= 64 and not half)
        if layer.up == 2 and fuse and ops.upconv_fir_supported(x, wt, cout, batch):
            # the whole up-sampling layer in one pass: the raw transposed-conv result stays on the chip (csrc/upconv_fir.hip)
            def fused_layer():
                return self._timed(key + "_upfir", flops, ops.upconv_fir, x, wt, cout, k_styles, k_dcoef, noise, ns, layer.bias,
                                   "lrelu", cfg.lrelu_alpha, gain, conv_clamp, batch=batch, x_absmax=x_absmax,
                                   y_absmax=y_absmax, y_f16=half)
            out = self._timed(f"up_layer:{x.shape[3]}->{cout}@{2 * x.shape[1]}:fused", flops, fused_layer)
        elif layer.up == 2:
            def two_kernels():
                yt = self._timed(key + "_up", flops, ops.modconv, x, wt, cout, ops.CONVT3X3_UP2, styles=k_styles, batch=batch,
                                 x_absmax=x_absmax, y_f16=half, x_parts=x_parts)
                return ops.upfir_epilogue(yt, k_dcoef, noise, ns, layer.bias, "lrelu", cfg.lrelu_alpha, gain, conv_clamp,
                                          y_absmax=y_absmax)
            # (bench.py `roofline.up_layers`: the LAYER — transposed-conv GEMM + FIR epilogue — per shape)
            out = self._timed(f"up_layer:{x.shape[3]}->{cout}@{2 * x.shape[1]}", flops, two_kernels)
        else:
            # rgb = (toRGB weight [3, Cout], toRGB styles [B, Cout]): form the block's toRGB sums in this conv's epilogue
            # when the kernel can (ops.fused_torgb_supported); the caller finishes them with ops.torgb_finish
            rgb_w = None
            if rgb is not None and ops.fused_torgb_supported(x, wt, cout, batch):
                rgb_w = (rgb[1][:, None, :] * rgb[0][None]).contiguous()
            out = self._timed(key, flops, ops.modconv, x, wt, cout, ops.CONV3X3, styles=k_styles, dcoef=k_dcoef,
                              noise=noise, noise_strength=ns, bias=layer.bias, act="lrelu", alpha=cfg.lrelu_alpha,
                              gain=gain, clamp=conv_clamp, batch=batch, x_absmax=x_absmax, y_absmax=y_absmax,
                              rgb_w=rgb_w, y_f16=half, x_parts=x_parts,
                              store_y=keep_out or rgb_w is None or tape is not None or y_absmax is not None)
            if rgb_w is not None:
                out, self._rgb_part = out
        rec = None
        if tape is not None:
            rec = dict(layer=layer, x=x, styles=styles, dcoef=dcoef, out=out, row=row, up=layer.up, wsq=wsq,
                       producer=dict(dcoef=dcoef, bias=layer.bias, noise=noise_raw, noise_strength=ns if ns_dev is None else 0.0,
                                     noise_strength_dev=ns_dev, act="lrelu", alpha=cfg.lrelu_alpha, gain=gain, clamp=conv_clamp))
        return out, rec

    def _block(self, x, img, blk: _SynthesisBlock, ws, rows, batch, noise_mode, conv_clamp, small_rgb, last, tape,
               absmax=None, half=False, keep_out=True):
        """keep_out=False: nobody reads this block's feature output x (the last super-resolution block): when its toRGB
        rides in conv1's epilogue and nothing is recorded for a backward pass, conv1 does not store its activation (x is
        then returned as None).
        rows = the ws row of (conv0,) conv1, torgb.  absmax = (slots of the block input | None, slots for conv0's
        output, slots for conv1's output) when the chain is unclamped (fp16 range tracking), else None."""
        rec = dict(conv0=None, first=blk.in_channels == 0, img_in=img, const=getattr(blk, "const", None))
        am_in, am0, am1 = absmax if absmax is not None else (None, None, None)
        tr = blk.torgb
        cin = tr.weight.shape[1]
        row = rows[-1]
        pre = self._styles.pop(id(tr), None) if self._styles else None
        styles = pre[0] if pre is not None else ops.styles_demod(ws[:, row], tr.affine.weight, tr.affine.bias, None,
                                                                 1.0 / math.sqrt(cin))[0]
        # a 3-channel toRGB (super-resolution blocks) rides in the epilogue of conv1 where the kernel allows it
        rgb = (tr.weight.detach().reshape(tr.weight.shape[0], cin), styles) if (small_rgb and tr.weight.shape[0] <= 3) else None
        self._rgb_part = None
        if blk.in_channels == 0:
            x, rec["conv1"] = self._layer(self._const(blk.const), blk.conv1, ws[:, rows[0]], rows[0], batch,
                                          noise_mode, conv_clamp, tape, None, am1, rgb)
        else:
            x, rec["conv0"] = self._layer(x, blk.conv0, ws[:, rows[0]], rows[0], batch, noise_mode, conv_clamp, tape,
                                          am_in, am0, None, half)
            x, rec["conv1"] = self._layer(x, blk.conv1, ws[:, rows[1]], rows[1], batch, noise_mode, conv_clamp, tape,
                                          am0, am1, rgb, half, keep_out)
        y = y_pre = None
        if small_rgb:
            if tape is not None and conv_clamp is not None:
                y_pre = torch.empty(batch, tr.weight.shape[0], x.shape[1], x.shape[2], device=x.device)
            if self._rgb_part is not None:
                img = ops.torgb_finish(self._rgb_part, tr.bias, img, conv_clamp, y_pre)
                self._rgb_part = None
            else:
                img = ops.torgb_small(x.float() if half else x, tr.weight.detach().reshape(tr.weight.shape[0], cin),
                                      styles, tr.bias, img, conv_clamp, y_pre)
        else:
            wt = self._gemm_image(tr.weight)          # (no demodulation: no wsq)
            side = self.side_stream(batch, x.device)
            main = torch.cuda.current_stream(x.device) if side is not None else None
            if side is not None:
                side.wait_stream(main)               # conv1's output (and everything before it) is ordered before the side chain
                for t in (x, styles):                # main-stream tensors read on the side stream: not to be recycled under it
                    t.record_stream(side)
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                # the block that writes the tri-planes also publishes max |planes|: the bound the ray marcher's 16-bit
                # decoder scales its operands by (ops.raymarch planes_absmax)
                self._planes_absmax = ops.absmax_slots(1, x.device)[0] if last else None
                if conv_clamp is None and x.shape[0] == batch and ops.torgb_skip_supported(x, wt, tr.weight.shape[0]):
                    # toRGB and the skip connection in one streaming pass (the toRGB output is never stored)
                    img = ops.torgb_skip(x, wt, tr.weight.shape[0], styles, tr.bias, img, plane_major=last, x_absmax=am1,
                                         out_absmax=self._planes_absmax)
                else:
                    y = ops.modconv(x, wt, tr.weight.shape[0], ops.CONV1X1, styles=styles, bias=tr.bias, act="linear",
                                    gain=1.0, clamp=conv_clamp, batch=batch, x_absmax=am1)
                    img = ops.skip_upsample_add(img, y, plane_major=last, out_absmax=self._planes_absmax)
            if side is not None and last:
                # join: the tri-planes (and their |max| slots) are consumed on the launch stream from here on
                main.wait_stream(side)
                img.record_stream(main)
                self._planes_absmax.record_stream(main)
        if tape is not None:
            rec["rgb"] = dict(torgb=tr, x=x, styles=styles, row=row, small=small_rgb, clamp=conv_clamp,
                              y=y if conv_clamp is not None else None, y_pre=y_pre)
            tape.append(rec)
        return x, img

    def _precompute_styles(self, ws: torch.Tensor, blocks_rows) -> None:
        """Styles (and demodulation coefficients) of every layer of ``blocks_rows`` = [(block, rows)] in two
        launches; `_layer` / `_block` pick them up by layer identity."""
        cfg = self.cfg
        items, keys = [], []
        for blk, rows in blocks_rows:
            convs = [blk.conv1] if blk.in_channels == 0 else [blk.conv0, blk.conv1]
            for layer, row in zip(convs, rows):
                items.append((ws[:, row], layer.affine.weight, layer.affine.bias, self._prepared(layer.weight)[1], 1.0,
                              cfg.demod_eps))
                keys.append(id(layer))
            tr = blk.torgb
            items.append((ws[:, rows[-1]], tr.affine.weight, tr.affine.bias, None, 1.0 / math.sqrt(tr.weight.shape[1]),
                          cfg.demod_eps))
            keys.append(id(tr))
        self._styles = {}
        for i in range(0, len(items), 32):
            for k, v in zip(keys[i:i + 32], ops.styles_demod_batch(items[i:i + 32])):
                self._styles[k] = v

    # ----------------------------------------------------------------- public API
    def backbone_planes(self, ws: torch.Tensor, tape=None) -> torch.Tensor:
        """ws [B, num_ws, 512] → tri-plane volume [B, 3, R, R, 32] (plane-major, channels-last)."""
        if not getattr(self, "_in_forward", False):     # called directly (not through synthesis): the derived images follow the weights
            self._refresh_tuned(tape is not None)
        cfg = self.cfg
        syn = self.backbone.synthesis
        b = ws.shape[0]
        x = img = None
        idx = 0
        plan = []
        for res in cfg.block_resolutions:
            n_conv = 1 if res == 4 else 2
            plan.append((getattr(syn, f"b{res}"), list(range(idx, idx + n_conv + 1))))
            idx += n_conv
        self._precompute_styles(ws, plan)
        # fp16 range tracking: the backbone has no clamp (conv_clamp None), so every conv publishes max |out| and the
        # next fp16-kind GEMM normalises its operand with it (one memset for all layers; row 2k / 2k+1 = conv0 / conv1
        # of block k).  Kept for `f16_range_report()`.
        track = cfg.backbone_conv_clamp is None and (self._conv_precision in ("f16x3", "f16x2", "f16"))
        am = ops.absmax_slots(2 * len(cfg.block_resolutions), ws.device) if track else None
        self._absmax = (am, [f"b{res}.{c}" for res in cfg.block_resolutions for c in ("conv0", "conv1")]) if track else None
        idx = 0
        prev = None
        for k, res in enumerate(cfg.block_resolutions):
            blk = getattr(syn, f"b{res}")
            n_conv = 1 if res == 4 else 2
            rows = list(range(idx, idx + n_conv + 1))
            x, img = self._block(x, img, blk, ws, rows, b, cfg.backbone_noise_mode, cfg.backbone_conv_clamp, False,
                                 res == cfg.plane_resolution, tape,
                                 (prev, am[2 * k], am[2 * k + 1]) if track else None)
            prev = am[2 * k + 1] if track else None
            idx += n_conv
        side = self._side_streams.get(ws.device.index if ws.device.index is not None else torch.cuda.current_device())
        if side is not None and not torch.cuda.is_current_stream_capturing():
            # join of the image side chain, whatever block ended it (not inside a HIP-graph capture: side_stream() declines there,
            # and waiting on a stream outside the capture would invalidate it — ADVICE r4)
            torch.cuda.current_stream(ws.device).wait_stream(side)
        return img

    def f16_range_report(self) -> Optional[Dict[str, float]]:
        """max |activation| of every backbone layer output of the LAST synthesis (host sync), or None when the conv
        precision has no fp16 parts / the backbone is clamped.  Values beyond 65504 are handled (the consumer GEMM
        scales by an exact power of two, hfagp.h `x_absmax`); the report shows how far a checkpoint is from the cliff
        that an un-tracked fp16 split would have."""
        if getattr(self, "_absmax", None) is None:
            return None
        am, names = self._absmax
        vals = am.max(dim=1).values.tolist()
        return {n: v for n, v in zip(names, vals) if not n.startswith("b4.conv0")}

    def _render_args(self, c: torch.Tensor):
        cfg = self.cfg
        net = self.decoder.net
        hit = getattr(self, "_cam_cache", None)       # the backward pass of a step asks again with the forward's tensor
        if hit is None or hit[0] is not c or hit[1] != c._version:
            hit = self._cam_cache = (c, c._version, c[:, :16].contiguous(), c[:, 16:25].contiguous())
        return dict(cam2world=hit[2], intrinsics=hit[3],
                    dec_w0=net["0"].weight, dec_b0=net["0"].bias, dec_w1=net["2"].weight, dec_b1=net["2"].bias,
                    res=cfg.neural_rendering_resolution, ray_start=cfg.ray_start, ray_end=cfg.ray_end,
                    box_warp=cfg.box_warp, decoder_lr_mul=cfg.decoder_lr_mul,
                    plane_axes=0 if cfg.plane_axes == "eg3d_original" else 1, white_back=cfg.white_back,
                    decoder_precision=cfg.decoder_precision)

    def _uniforms(self, b: int, dev, u_strat, u_imp):
        cfg = self.cfg
        r = cfg.neural_rendering_resolution ** 2
        if u_strat is None:
            u_strat = torch.rand(b, r, cfg.depth_resolution, device=dev)
        if u_imp is None:
            u_imp = torch.rand(b * r, cfg.depth_resolution_importance, device=dev)
        return u_strat.reshape(b, r, -1).contiguous(), u_imp.contiguous()

    def render(self, planes: torch.Tensor, c: torch.Tensor, u_strat=None, u_imp=None, planes_absmax=None, state=None):
        """state (ops.raymarch_state): filled with the per-sample results the backward pass of this step will need."""
        cfg = self.cfg
        b = planes.shape[0]
        r = cfg.neural_rendering_resolution ** 2
        u_strat, u_imp = self._uniforms(b, planes.device, u_strat, u_imp)
        # algorithmic bytes per frame: samples * 3 planes * 4 taps * 32 ch * 4 B
        # + outputs R*(32+1+1)*4 + uniforms R*(Sc+Sf)*4
        s_tot = cfg.depth_resolution + cfg.depth_resolution_importance
        nbytes = float(b) * r * (s_tot * 3 * 4 * 32 * 4 + 34 * 4 + s_tot * 4)
        return self._timed("raymarch", nbytes, ops.raymarch, planes, u_strat=u_strat, u_imp=u_imp,
                           planes_absmax=planes_absmax, state=state, **self._render_args(c))

    def superres(self, rgb_raw: torch.Tensor, feat_img: torch.Tensor, ws: torch.Tensor, tape=None) -> torch.Tensor:
        if not getattr(self, "_in_forward", False):     # called directly: see backbone_planes
            self._refresh_tuned(tape is not None)
        cfg = self.cfg
        b = ws.shape[0]
        last = cfg.num_ws - 1
        sr = self.superresolution
        self._precompute_styles(ws, [(sr.block0, [last] * 3), (sr.block1, [last] * 3)])
        half = self._sr_half(b, feat_img.shape[1], tape)
        x, rgb = self._block(feat_img, rgb_raw, sr.block0, ws, [last] * 3, b, cfg.sr_noise_mode, cfg.sr_conv_clamp,
                             True, False, tape, None, half)
        x, rgb = self._block(x, rgb, sr.block1, ws, [last] * 3, b, cfg.sr_noise_mode, cfg.sr_conv_clamp, True, False,
                             tape, None, half, keep_out=False)
        return rgb

    def _sr_half(self, batch: int, res: int, tape) -> bool:
        """fp16 STORAGE of the super-resolution activations (cfg.sr_storage = "f16"): EG3D's own arrangement for these two
        blocks (sr_num_fp16_res = 4: fp16 tensors between the layers, clamp 256).  Taken only where the arithmetic is
        the single-pass fp16 one (sr_conv_precision "f16"), nothing is recorded for a backward pass, and every conv of
        the two blocks is one the kernels can run that way (ops.f16_storage_supported)."""
        cfg = self.cfg
        if self.sr_storage != "f16" or self._sr_conv_precision != "f16" or tape is not None:
            return False
        if cfg.sr_conv_clamp is None:
            return False                  # fp16 storage relies on the clamp (|x| <= 256) for its range
        c0, c1 = cfg.sr_channels
        ok = ops.f16_storage_supported
        return (ok(res, res, cfg.plane_channels, c0, batch) and ok(2 * res, 2 * res, c0, c0, batch) and
                ok(2 * res, 2 * res, c0, c1, batch) and ok(4 * res, 4 * res, c1, c1, batch))

    def _check_inputs(self, ws, c, noise_mode):
        cfg = self.cfg
        if not ws.is_cuda:
            raise RuntimeError("TriPlaneGenerator.synthesis: the MI355X path needs CUDA/ROCm tensors; "
                               "there is no CPU fallback (oracle/ is test infrastructure only)")
        if noise_mode != "const":
            raise NotImplementedError("HFA-GP always passes noise_mode='const' (headnerf.py:112)")
        if ws.device.index != torch.cuda.current_device():
            raise RuntimeError(f"TriPlaneGenerator.synthesis: tensors live on {ws.device} but the current device is "
                               f"cuda:{torch.cuda.current_device()}; the kernels are enqueued on the CURRENT device's "
                               f"stream — wrap the call in `with torch.cuda.device(ws.device):`")
        if c.requires_grad and torch.is_grad_enabled():
            raise RuntimeError("TriPlaneGenerator.synthesis: the camera label `c` requires grad, but the renderer "
                               "has no camera gradient (HFA-GP never optimises the pose through the generator); "
                               "detach it explicitly")
        if ws.shape[1:] != (cfg.num_ws, cfg.w_dim) or c.shape[1] != cfg.c_dim:
            raise ValueError(f"expected ws [B,{cfg.num_ws},{cfg.w_dim}] and c [B,{cfg.c_dim}], got "
                             f"{tuple(ws.shape)} and {tuple(c.shape)}")

    def _forward_impl(self, ws, c, u_strat, u_imp, tape):
        """ws, c: detached contiguous fp32 CUDA tensors.  Returns image, image_raw, depth, planes, feat_img;
        when `tape` is a dict it is filled with everything the backward pass needs."""
        self._refresh_tuned(tape is not None)
        self._in_forward = True          # (backbone_planes / superres below must not rebuild the images a second time)
        try:
            return self._forward_body(ws, c, u_strat, u_imp, tape)
        finally:
            self._in_forward = False

    def _forward_body(self, ws, c, u_strat, u_imp, tape):
        cfg = self.cfg
        b = ws.shape[0]
        res = cfg.neural_rendering_resolution
        bb_tape = [] if tape is not None else None
        sr_tape = [] if tape is not None else None
        planes = self.backbone_planes(ws, bb_tape)
        u_strat, u_imp = self._uniforms(b, ws.device, u_strat, u_imp)
        pam = getattr(self, "_planes_absmax", None)
        # a step that will be differentiated keeps the ray marcher's per-sample results (220 MB per frame at 128^2 rays x 96
        # samples; skipped above 8 GB): the compositing adjoint then reads them instead of marching every ray again
        ray_state = None
        if tape is not None and b * res * res * (cfg.depth_resolution + cfg.depth_resolution_importance) * 140 <= (8 << 30):
            ray_state = ops.raymarch_state(b, res, cfg.depth_resolution, cfg.depth_resolution_importance, ws.device)
        feat, depth, wsum, tmm = self.render(planes, c, u_strat, u_imp, planes_absmax=pam, state=ray_state)
        # MipRayMarcher2 clamps the expected depth to the GLOBAL min/max sample depth of the batch
        depth = ops.depth_clamp_(depth, tmm)
        feat_img = feat.view(b, res, res, 32)                             # channels-last
        rgb_raw = feat_img[..., :3].permute(0, 3, 1, 2).contiguous()      # NCHW, 'image_raw'
        img = self.superres(rgb_raw, feat_img, ws, sr_tape)
        if tape is not None:
            tape.update(backbone=bb_tape, sr=sr_tape, planes=planes, c=c, u_strat=u_strat, u_imp=u_imp,
                        feat_img=feat_img, batch=b, planes_absmax=pam, ray_state=ray_state)
        return img, rgb_raw, depth.view(b, 1, res, res), planes, feat_img

    def synthesis(self, ws: torch.Tensor, c: torch.Tensor, noise_mode: str = "const",
                  u_strat: Optional[torch.Tensor] = None, u_imp: Optional[torch.Tensor] = None,
                  return_planes: bool = False, **_unused) -> Dict[str, torch.Tensor]:
        """Drop-in for EG3D's TriPlaneGenerator.synthesis.  Differentiable w.r.t. `ws` (the latent-basis fitting of
        HFA-GP) and w.r.t. the generator parameters that require grad (after `tune_generator()`)."""
        self._check_inputs(ws, c, noise_mode)
        if ws.shape[0] == 0:          # empty batch (ragged last batch of a frame shard): nothing to launch
            cfg, dev = self.cfg, ws.device
            r = cfg.neural_rendering_resolution
            out = {"image": torch.zeros(0, cfg.img_channels, cfg.img_resolution, cfg.img_resolution, device=dev),
                   "image_raw": torch.zeros(0, 3, r, r, device=dev), "image_depth": torch.zeros(0, 1, r, r, device=dev)}
            out["image"] = out["image"] + 0.0 * ws.sum()          # keeps the autograd edge to ws
            return out
        params = [p for n, p in self.named_parameters() if not n.startswith("backbone.mapping.")]
        need_grad = torch.is_grad_enabled() and (ws.requires_grad or any(p.requires_grad for p in params))
        if need_grad:
            from .autograd import SynthesisFn
            img, rgb_raw, depth = SynthesisFn.apply(ws, c.detach(), u_strat, u_imp, self, *params)
            out = {"image": img, "image_raw": rgb_raw, "image_depth": depth}
            if return_planes:
                out["planes"], out["feature_image"] = self._last_extras
            self._last_extras = None
            return out
        with torch.no_grad():
            img, rgb_raw, depth, planes, feat_img = self._forward_impl(
                ws.detach().float().contiguous(), c.detach().float().contiguous(), u_strat, u_imp, None)
        out = {"image": img, "image_raw": rgb_raw, "image_depth": depth}
        if return_planes:
            out["planes"] = planes
            out["feature_image"] = feat_img
        return out

    @torch.no_grad()
    def mapping(self, z: torch.Tensor, c: torch.Tensor, truncation_psi: float = 1.0) -> torch.Tensor:
        """EG3D MappingNetwork.forward (z, c) -> ws [B, num_ws, 512].  HFA-GP never calls it (its ws come from the
        latent basis, headnerf.py:81-102); provided for unconditional sampling.  Inference only."""
        cfg = self.cfg
        mp = self.backbone.mapping

        def norm2(x):       # normalize_2nd_moment: tiny host-side glue
            return x * (x.square().mean(1, keepdim=True) + 1e-8).rsqrt()
        x = norm2(z.float().contiguous())
        y = norm2(ops.fully_connected(c.float().contiguous(), mp.embed.weight, mp.embed.bias))
        x = torch.cat([x, y], 1).contiguous()
        for i in range(cfg.mapping_layers):
            fc = getattr(mp, f"fc{i}")
            x = ops.fully_connected(x, fc.weight, fc.bias, cfg.mapping_lr_mul, act="lrelu", alpha=0.2)
        ws = x[:, None].repeat(1, cfg.num_ws, 1)
        if truncation_psi != 1.0:
            ws = mp.w_avg.lerp(ws, truncation_psi)
        return ws

    def forward(self, *args, **kwargs):
        raise NotImplementedError("HFA-GP only calls generator.synthesis (headnerf.py:112); mapping is in mapping()")


def load_G_official(args=None, device="cuda", cfg: Optional[GeneratorConfig] = None, seed: int = 0,
                    weights: Optional[str] = None) -> TriPlaneGenerator:
    """Counterpart of headnerf.py:31-38.  The EG3D pickle is not shipped with the reference, so the
    generator is random-initialised from ``seed`` (EG3D init) or loaded from a safetensors file with
    EG3D key names; it comes back frozen, as the reference does (``requires_grad_(False)``)."""
    if cfg is None and getattr(args, "generator_preset", None):
        from .config import PRESETS
        cfg = PRESETS[args.generator_preset]()
    seed = getattr(args, "generator_seed", seed)
    weights = weights or getattr(args, "generator_weights", None)
    g = TriPlaneGenerator(cfg, seed=seed)
    if weights is not None:
        from safetensors.torch import load_file
        g.load_state_dict(load_file(weights), strict=True)
    return g.requires_grad_(False).to(device)

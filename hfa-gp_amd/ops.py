"""Operator layer: the EG3D op API of the hot path, backed by libhfagp_hip.so.

Mirrors (names and argument meaning) the operators the reference's generator
reaches below ``generator.synthesis`` (headnerf.py:112): ``bias_act``,
``upfirdn2d`` / ``upsample2d``, ``modulated_conv2d`` (+ fused ``bias_act``),
and the renderer.  Every function takes CUDA (ROCm) fp32 tensors, enqueues on
``torch.cuda.current_stream()`` and never falls back to PyTorch math.
Activations are channels-last ``[B, H, W, C]`` unless a name says ``nchw``.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Tuple

import torch

from . import _lib as L

CONV3X3, CONVT3X3_UP2, CONV1X1, CONV3X3_BWD, CONVS2_BWD = 0, 1, 2, 3, 4
ACT_LINEAR, ACT_LRELU = 0, 1
_ACT = {"linear": ACT_LINEAR, "lrelu": ACT_LRELU}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream() -> int:
    """hipStream_t of torch's current stream on the current device.  The raw accessor is ~20x cheaper than
    ``torch.cuda.current_stream()`` (which builds a Stream object): it is called once per launch, ~350 times per
    fitting step."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()


def _chk(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a CUDA/ROCm tensor (the HIP path has no CPU fallback)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    return t


# ----------------------------------------------------------------------------- weights
def weight_prep(weight: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """weight [Cout, Cin, k, k] → (wt [k*k, Cin/4, Cout, 4], wsq [Cout, Cin])."""
    _chk(weight, "weight")
    co, ci, kh, kw = weight.shape
    wt = torch.empty(kh * kw, ci // 4, co, 4, device=weight.device, dtype=torch.float32)
    wsq = torch.empty(co, ci, device=weight.device, dtype=torch.float32)
    L.check(L.lib().hfagp_weight_prep(_ptr(weight), _ptr(wt), _ptr(wsq), co, ci, kh * kw, _stream()), "weight_prep")
    return wt, wsq


PREC_F32, PREC_BF16X3, PREC_BF16X6, PREC_F16, PREC_F16X3, PREC_F16X2 = 0, 1, 2, 3, 4, 5
# "f16x2" (TF32 class, NOT the default): the f16x3 weight image against activations rounded to one fp16 part — 2 MFMAs / product
PRECISIONS = {"fp32": PREC_F32, "bf16x3": PREC_BF16X3, "bf16x6": PREC_BF16X6, "f16": PREC_F16, "f16x3": PREC_F16X3,
              "f16x2": PREC_F16X2}
NPARTS = {"fp32": 0, "f16": 1, "bf16x3": 2, "bf16x6": 3, "f16x3": 2, "f16x2": 2}   # parts of the 16-bit weight image (0: fp32 image)
_IMAGE_DTYPE = {"f16": torch.float16, "f16x3": torch.float16, "f16x2": torch.float16, "bf16x3": torch.bfloat16,
                "bf16x6": torch.bfloat16}


def split_supported(cin: int, cout: int, up: bool = False) -> bool:
    """Shapes the 16-bit conv paths take (hfagp.h: Cin % 16 == 0 and Cout % 128 == 0, or Cout % 128 >= 96 for the
    non-upsampling modes: the 96-channel toRGB on a 128-wide tile whose last 32 columns are discarded)."""
    return cin % 16 == 0 and (cout % 128 == 0 or (cout % 128 >= 96 and not up))


def weight_prep_prec(weight: torch.Tensor, precision: str) -> torch.Tensor:
    """weight [Cout, Cin, k, k] fp32 → B-operand image [parts, k*k, Cin/8, Cout, 8] of a 16-bit conv precision:
    'bf16x3' / 'bf16x6' = 2 / 3 bfloat16 parts, 'f16x3' = 2 float16 parts (each part the round-to-nearest value of
    the residual left by the parts before it), 'f16' = one float16 part."""
    _chk(weight, "weight")
    co, ci, kh, kw = weight.shape
    nparts = NPARTS[precision]
    # + 512 B behind the image: with Cout % 128 != 0 the 128-wide tile reads one partial row past it (hfagp.h).  Only readable
    # memory is needed there — the columns fed from it are never stored — so nothing is cleared (the RGB fitting step re-builds
    # 21 images per step: clearing them was 0.24 ms of fill kernels).  INVARIANT the kernels keep: no consumer reduces over the
    # discarded columns (each output column of the GEMM depends on its own B column only; the fused toRGB sums and the absmax
    # reduction run over stored columns) — tests/test_gpu_parity.py::test_torgb96_on_padded_split_tile runs with the pad poisoned.
    numel = nparts * kh * kw * (ci // 8) * co * 8
    flat = torch.empty(numel + 256, device=weight.device, dtype=_IMAGE_DTYPE[precision])
    wb = flat[:numel].view(nparts, kh * kw, ci // 8, co, 8)
    L.check(L.lib().hfagp_weight_prep_prec(_ptr(weight), wb.data_ptr(), co, ci, kh * kw, PRECISIONS[precision], _stream()),
            "weight_prep_prec")
    return wb


def weight_prep_batch_supported(weight: torch.Tensor) -> bool:
    co, ci, kh, kw = weight.shape
    return co % 32 == 0 and ci % 32 == 0 and kh == kw and kh in (1, 3)


def weight_prep_batch(items):
    """items: [(weight [Cout,Cin,k,k], precision | None, precision_t | None, want_wsq)] -> [(image | None, image_t | None, wsq | None)]
    in ONE launch per 48 weights (hfagp_weight_prep_batch): the forward image, the image of the Cin/Cout transpose and wsq of every
    weight from a single read of it.  All outputs of a call live in three flat buffers (one allocation each)."""
    metas, n16 = [], {torch.float16: 0, torch.bfloat16: 0}
    nsq = 0
    for w, prec, prec_t, want_wsq in items:
        _chk(w, "weight")
        co, ci, kh, kw = w.shape
        sizes = []
        for pr in (prec, prec_t):
            if pr is None:
                sizes.append(None)
            else:
                dt = _IMAGE_DTYPE[pr]
                numel = NPARTS[pr] * kh * kw * ci * co            # (same element count for the transpose)
                sizes.append((dt, n16[dt], numel))
                n16[dt] += (numel + 256 + 7) // 8 * 8            # + the 512-byte readable pad behind an image (see weight_prep_prec)
        metas.append((sizes, nsq if want_wsq else None))
        if want_wsq:
            nsq += co * ci
    dev = items[0][0].device
    flat = {dt: torch.empty(n, device=dev, dtype=dt) for dt, n in n16.items() if n}
    sq = torch.empty(nsq, device=dev, dtype=torch.float32) if nsq else None
    out, arr = [], (L.WeightPrepItem * len(items))()
    for k, ((w, prec, prec_t, want_wsq), (sizes, sq_off)) in enumerate(zip(items, metas)):
        co, ci, kh, kw = w.shape
        views = []
        for pr, sz, shape in ((prec, sizes[0], (kh * kw, ci // 8, co, 8)), (prec_t, sizes[1], (kh * kw, co // 8, ci, 8))):
            if sz is None:
                views.append(None)
            else:
                dt, off, numel = sz
                views.append(flat[dt][off: off + numel].view(NPARTS[pr], *shape))
        wsq = sq[sq_off: sq_off + co * ci].view(co, ci) if want_wsq else None
        a = arr[k]
        a.weight = _ptr(w)
        a.image = views[0].data_ptr() if views[0] is not None else None
        a.image_t = views[1].data_ptr() if views[1] is not None else None
        a.wsq = _ptr(wsq)
        a.Cout, a.Cin, a.taps = co, ci, kh * kw
        a.precision = PRECISIONS[prec] if prec is not None else 0
        a.precision_t = PRECISIONS[prec_t] if prec_t is not None else 0
        out.append((views[0], views[1], wsq))
    for k0 in range(0, len(items), 48):
        n = min(48, len(items) - k0)
        sub = (L.WeightPrepItem * n).from_buffer(arr, k0 * C.sizeof(L.WeightPrepItem))
        L.check(L.lib().hfagp_weight_prep_batch(sub, n, _stream()), "weight_prep_batch")
    return out


def weight_prep_split(weight: torch.Tensor, nparts: int) -> torch.Tensor:
    """`weight_prep_prec` by part count: 1 = 'f16', 2 = 'bf16x3', 3 = 'bf16x6'."""
    return weight_prep_prec(weight, {1: "f16", 2: "bf16x3", 3: "bf16x6"}[nparts])


def styles_demod(w: torch.Tensor, affine_w: torch.Tensor, affine_b: torch.Tensor,
                 wsq: Optional[torch.Tensor], style_gain: float = 1.0, eps: float = 1e-8):
    """w [B, w_dim] (may be a strided row view of ws) → styles [B, Cin], dcoef [B, Cout] | None."""
    if w.dtype != torch.float32 or not w.is_cuda or w.stride(-1) != 1:
        raise RuntimeError("styles_demod: w must be a CUDA fp32 tensor with unit inner stride")
    b, wd = w.shape
    cin = affine_w.shape[0]
    styles = torch.empty(b, cin, device=w.device, dtype=torch.float32)
    dcoef = None
    a = L.StyleArgs()
    a.w, a.affine_w, a.affine_b = _ptr(w), _ptr(_chk(affine_w, "affine_w")), _ptr(_chk(affine_b, "affine_b"))
    a.styles = _ptr(styles)
    a.B, a.w_dim, a.w_stride, a.Cin = b, wd, w.stride(0), cin
    a.style_gain, a.eps = style_gain, eps
    if wsq is not None:
        dcoef = torch.empty(b, wsq.shape[0], device=w.device, dtype=torch.float32)
        a.wsq, a.dcoef, a.Cout = _ptr(_chk(wsq, "wsq")), _ptr(dcoef), wsq.shape[0]
    L.check(L.lib().hfagp_style_fwd(C.byref(a), _stream()), "style_fwd")
    return styles, dcoef


def styles_demod_batch(items):
    """``items``: sequence of (w [B, w_dim] row view, affine_w, affine_b, wsq | None, style_gain, eps), at most 32.
    One launch for all the styles, one for all the demodulation coefficients.  Returns [(styles, dcoef | None)]."""
    n = len(items)
    arr = (L.StyleArgs * n)()
    out = []
    for i, (w, affine_w, affine_b, wsq, style_gain, eps) in enumerate(items):
        if w.dtype != torch.float32 or not w.is_cuda or w.stride(-1) != 1:
            raise RuntimeError("styles_demod_batch: w must be a CUDA fp32 tensor with unit inner stride")
        b, wd = w.shape
        cin = affine_w.shape[0]
        styles = torch.empty(b, cin, device=w.device, dtype=torch.float32)
        dcoef = None
        a = arr[i]
        a.w, a.affine_w, a.affine_b = _ptr(w), _ptr(_chk(affine_w, "affine_w")), _ptr(_chk(affine_b, "affine_b"))
        a.styles = _ptr(styles)
        a.B, a.w_dim, a.w_stride, a.Cin = b, wd, w.stride(0), cin
        a.style_gain, a.eps = style_gain, eps
        if wsq is not None:
            dcoef = torch.empty(b, wsq.shape[0], device=w.device, dtype=torch.float32)
            a.wsq, a.dcoef, a.Cout = _ptr(_chk(wsq, "wsq")), _ptr(dcoef), wsq.shape[0]
        out.append((styles, dcoef))
    L.check(L.lib().hfagp_style_batch_fwd(arr, n, _stream()), "style_batch_fwd")
    return out


def qr_gram(gram: torch.Tensor, top: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """gram = A^T A [n, n], top = A[:n, :n] → (R with LAPACK's signs, R^-1); Q = A @ R^-1 (n <= 64)."""
    n = gram.shape[0]
    r = torch.empty(n, n, device=gram.device, dtype=torch.float32)
    rinv = torch.empty(n, n, device=gram.device, dtype=torch.float32)
    L.check(L.lib().hfagp_qr_gram_fwd(_ptr(_chk(gram, "gram")), _ptr(_chk(top, "top")), _ptr(r), _ptr(rinv), n, _stream()),
            "qr_gram_fwd")
    return r, rinv


def qr_refine(gram: torch.Tensor, status: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """gram = Q1^T Q1 [n, n] → (upper Cholesky factor R2 with positive diagonal, R2^-1); status [2] (optional)
    receives (max |gram - I|, breakdown flag).  See include/hfagp.h."""
    n = gram.shape[0]
    r = torch.empty(n, n, device=gram.device, dtype=torch.float32)
    rinv = torch.empty(n, n, device=gram.device, dtype=torch.float32)
    L.check(L.lib().hfagp_qr_refine_fwd(_ptr(_chk(gram, "gram")), _ptr(r), _ptr(rinv), _ptr(status), n, _stream()),
            "qr_refine_fwd")
    return r, rinv


def tall_gram(x: torch.Tensor, y: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """scale * x^T y [n, n] for tall-skinny fp32 matrices [m, n <= 64], row- or column-major views (hfagp_tall_gram)."""
    m, n = x.shape
    for t in (x, y):
        if t.shape != (m, n) or t.dtype != torch.float32 or not t.is_cuda or 1 not in (t.stride(0), t.stride(1)):
            raise RuntimeError("tall_gram: operands must be fp32 device matrices of one shape with a unit stride")
    out = torch.empty(n, n, device=x.device, dtype=torch.float32)
    ws = torch.empty(L.lib().hfagp_tall_gram_workspace_bytes(m, n) // 4, device=x.device, dtype=torch.float32)
    L.check(L.lib().hfagp_tall_gram(x.data_ptr(), x.stride(0), x.stride(1), y.data_ptr(), y.stride(0), y.stride(1), _ptr(ws),
                                    _ptr(out), m, n, scale, _stream()), "tall_gram")
    return out


class TallSkinnyQR(torch.autograd.Function):
    """Q of the reduced QR of a tall-skinny fp32 CUDA matrix A [m, n <= 64] with torch.linalg.qr's (LAPACK's) sign
    convention; backward = the standard QR adjoint for dR = 0:
        dA = (dQ + Q S) R^-T,   X = triu(-Q^T dQ),   S = X + X^T - diag(X).
    Two passes: the Gram-matrix Householder kernel (Q1 = A R1^-1, orthogonality defect ~cond(A)^2 eps) and one
    Cholesky re-orthogonalisation of Q1 (Q = Q1 R2^-1, defect O(eps) while cond(A)^2 eps < ~0.3); R^-1 = R1^-1 R2^-1.
    `status` (device float[2], optional) receives the defect of pass 1 and the breakdown flag (ops.qr_refine)."""

    @staticmethod
    def forward(ctx, a: torch.Tensor, status: Optional[torch.Tensor] = None) -> torch.Tensor:
        n = a.shape[1]
        gram = tall_gram(a, a)
        _, rinv1 = qr_gram(gram, a[:n, :n].contiguous())
        q1 = a @ rinv1
        _, rinv2 = qr_refine(tall_gram(q1, q1), status)
        rinv = rinv1 @ rinv2
        q = q1 @ rinv2
        ctx.save_for_backward(q, rinv)
        return q

    @staticmethod
    def backward(ctx, gq: torch.Tensor):
        q, rinv = ctx.saved_tensors
        if gq.stride(0) != 1 and gq.stride(1) != 1:
            gq = gq.contiguous()
        x = torch.triu(tall_gram(q, gq, -1.0))
        s = x + x.T - torch.diag(torch.diagonal(x))
        return (gq + q @ s) @ rinv.T, None


def fully_connected(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], lr_mul: float = 1.0,
                    act: str = "linear", alpha: float = 0.2, gain: Optional[float] = None) -> torch.Tensor:
    """EG3D FullyConnectedLayer: act((x @ (W*lr_mul/sqrt(in)).T + b*lr_mul)) * gain."""
    _chk(x, "x")
    b, fin = x.shape
    fout = weight.shape[0]
    if gain is None:
        gain = math.sqrt(2.0) if act == "lrelu" else 1.0
    y = torch.empty(b, fout, device=x.device, dtype=torch.float32)
    L.check(L.lib().hfagp_fc_fwd(_ptr(x), _ptr(_chk(weight, "weight")), _ptr(bias), _ptr(y), b, fin, fout, lr_mul,
                                 _ACT[act], alpha, gain, _stream()), "fc_fwd")
    return y


# ----------------------------------------------------------------------------- modulated conv
def absmax_slots(n: int, device) -> torch.Tensor:
    """[n, 64 * 32] zeroed slot buffers (64 slots, one 128-byte line each) for the fp16 range tracking of n tensors (row i: max |.| of tensor i = row.max())."""
    return torch.zeros(n, L.ABSMAX_FLOATS, device=device, dtype=torch.float32)


def modconv(x: torch.Tensor, wt: torch.Tensor, cout: int, mode: int, styles: Optional[torch.Tensor] = None,
            dcoef: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
            noise_strength: float = 0.0, bias: Optional[torch.Tensor] = None, act: str = "linear",
            alpha: float = 0.2, gain: float = 1.0, clamp: Optional[float] = None, batch: Optional[int] = None,
            ksplit: int = 0, x_absmax: Optional[torch.Tensor] = None,
            y_absmax: Optional[torch.Tensor] = None, rgb_w: Optional[torch.Tensor] = None, y_f16: bool = False,
            store_y: bool = True, x_parts: int = 0):
    """x [B|1, H, W, Cin] channels-last.  mode CONV3X3 / CONV1X1: fused epilogue, returns [B,H,W,Cout];
    mode CONVT3X3_UP2: returns the RAW transposed-conv result [B, 2H+1, 2W+1, Cout].
    ``wt`` from :func:`weight_prep` (fp32, exact MFMA) or :func:`weight_prep_split` (bfloat16 parts: the
    split-bf16 MFMA path, BF16X3 for 2 parts, BF16X6 for 3; float16, 1 part: the single-pass fp16 path).
    ``x_absmax`` / ``y_absmax`` ([64 x 32] fp32, `absmax_slots`): fp16 range tracking — max |x| of the input as published by its
    producer (the fp16 kinds scale the operand by an exact power of two so nothing saturates) and the slot buffer that
    receives max |y| of a fused-epilogue output (include/hfagp.h).
    ``rgb_w`` [B, 3, Cout] (toRGB weight x its styles): fused toRGB — returns (y, rgb_part [parts, B, H, W, 4]) for
    `torgb_finish`; only where `fused_torgb_supported` says so.  ``store_y=False`` (with rgb_w): y is not written and None
    is returned in its place (last super-resolution layer of a forward-only call).
    ``x_parts=1`` with the two-part float16 image: precision 'f16x2' — the activations as ONE fp16 part against the 22-bit
    weights, two MFMAs per product (TF32 class; include/hfagp.h HFAGP_PREC_F16X2).
    fp16 STORAGE (single-pass fp16 weights only, `f16_storage_supported`): a float16 ``x`` is read as stored and
    ``y_f16`` writes the result as float16 (EG3D's fp16 super-resolution blocks keep their activations in fp16)."""
    x_f16 = x.dtype == torch.float16
    if x_f16 or y_f16:
        if not (wt.dtype == torch.float16 and wt.dim() == 5 and wt.shape[0] == 1):
            raise RuntimeError("modconv: fp16 storage (float16 x / y_f16) goes with the single-pass fp16 weight image")
        if not (x.is_cuda and x.is_contiguous()):
            raise RuntimeError("x: expected a contiguous CUDA/ROCm tensor")
    if not x_f16:
        _chk(x, "x")
    if mode == CONVS2_BWD:                      # x = four parity images [2,2,B,H+1,W+1,Cin]
        _, _, xb, h, w, cin = x.shape
        h, w = h - 1, w - 1
    else:
        xb, h, w, cin = x.shape
    b = batch if batch is not None else xb
    a = L.ModconvArgs()
    if wt.dtype in (torch.bfloat16, torch.float16):
        nparts = (2, 3) if wt.dtype == torch.bfloat16 else (1, 2)
        if not (wt.is_cuda and wt.is_contiguous() and wt.dim() == 5 and wt.shape[0] in nparts):
            raise RuntimeError("modconv: 16-bit weight images must come from weight_prep_prec")
        a.x, a.wt = x.data_ptr(), wt.data_ptr()
        if wt.dtype == torch.float16:
            a.precision = PREC_F16 if wt.shape[0] == 1 else PREC_F16X3
        else:
            a.precision = PREC_BF16X3 if wt.shape[0] == 2 else PREC_BF16X6
        if x_parts == 1 and a.precision != PREC_F16:
            if a.precision != PREC_F16X3:
                raise RuntimeError("modconv: x_parts=1 ('f16x2') goes with the two-part float16 weight image")
            a.precision = PREC_F16X2
    else:
        a.x, a.wt = _ptr(x), _ptr(_chk(wt, "wt"))
        a.precision = PREC_F32
    a.styles, a.dcoef, a.noise, a.bias = _ptr(styles), _ptr(dcoef), _ptr(noise), _ptr(bias)
    a.x_batch_stride = 0 if (xb == 1 and b > 1) else x.shape[-3] * x.shape[-2] * cin
    a.B, a.H, a.W, a.Cin, a.Cout = b, h, w, cin, cout
    a.mode, a.act, a.ksplit = mode, _ACT[act], ksplit
    a.noise_strength, a.alpha, a.gain = noise_strength, alpha, gain
    a.clamp = -1.0 if clamp is None else float(clamp)
    a.x_absmax, a.y_absmax = _ptr(x_absmax), _ptr(y_absmax)
    a.x_f16, a.y_f16 = int(x_f16), int(y_f16)
    ydt = torch.float16 if y_f16 else torch.float32
    if not store_y:
        # only the fused toRGB sums are wanted (`rgb_w`): the activation is not written at all
        if rgb_w is None or y_absmax is not None:
            raise RuntimeError("modconv: store_y=False needs rgb_w (the fused toRGB is then the only output) and no y_absmax")
        y = None
    elif mode == CONVT3X3_UP2:
        y = torch.empty(b, 2 * h + 1, 2 * w + 1, cout, device=x.device, dtype=ydt)
    else:
        y = torch.empty(b, h, w, cout, device=x.device, dtype=ydt)
    a.y = y.data_ptr() if y is not None else None
    if rgb_w is not None:
        a.rgb_w = _ptr(_chk(rgb_w, "rgb_w"))   # set BEFORE the workspace query: the kernel choice (small-image vs staged) reads it
    nbytes = L.lib().hfagp_modconv_workspace_bytes(C.byref(a))
    ws = None
    if nbytes:
        ws = torch.empty(nbytes // 4, device=x.device, dtype=torch.float32)
        a.workspace = _ptr(ws)
    part = None
    if rgb_w is not None:
        part = torch.empty(L.lib().hfagp_modconv_rgb_parts(C.byref(a)), b, h, w, 4, device=x.device, dtype=torch.float32)
        a.rgb_w, a.rgb_part = _ptr(_chk(rgb_w, "rgb_w")), _ptr(part)
    L.check(L.lib().hfagp_modconv_fwd(C.byref(a), _stream()), "modconv_fwd")
    return y if rgb_w is None else (y, part)


def f16_storage_supported(h: int, w: int, cin: int, cout: int, batch: int) -> bool:
    """Whether a 3x3 / up-sampling modconv on an h x w input grid can keep its activations in fp16 (`modconv` y_f16,
    float16 x): 128-channel output tiles and a launch the library does not split along K (its rule, modconv_plan.h: more than
    half a block per CU — 129 blocks of 8 x 16 positions x 128 channels — or fewer than four 16-channel K chunks)."""
    if cin % 16 != 0 or cout % 128 != 0:
        return False
    return cin < 64 or batch * ((h + 7) // 8) * ((w + 15) // 16) * (cout // 128) >= 129


def fused_torgb_supported(x: torch.Tensor, wt: torch.Tensor, cout: int, batch: int) -> bool:
    """Whether `modconv(..., mode=CONV3X3, rgb_w=...)` can form the toRGB sums in its epilogue: 16-bit weight image,
    Cout a multiple of 128, and a grid large enough that the library does not split K (then the epilogue lives in the
    reducer): the library's own rule, at least 129 blocks of 8 x 16 positions x 128 channels (modconv_plan.h)."""
    if wt.dtype == torch.float32 or cout % 128 != 0:
        return False
    h, w = x.shape[1], x.shape[2]
    return batch * ((h + 7) // 8) * ((w + 15) // 16) * (cout // 128) >= 129


def torgb_finish(part: torch.Tensor, bias: torch.Tensor, rgb_in: Optional[torch.Tensor], clamp: Optional[float],
                 y_pre: Optional[torch.Tensor] = None) -> torch.Tensor:
    """rgb_part of a fused-toRGB conv -> NCHW image: clamp(sum of parts + bias) + upsample2d(rgb_in)."""
    nparts, b, h, w, _ = part.shape
    cout = bias.shape[0]
    out = torch.empty(b, cout, h, w, device=part.device, dtype=torch.float32)
    a = L.TorgbFinishArgs()
    a.part, a.bias, a.rgb_out, a.y_pre = _ptr(part), _ptr(_chk(bias, "bias")), _ptr(out), _ptr(y_pre)
    a.rgb_in = _ptr(_chk(rgb_in, "rgb_in")) if rgb_in is not None else None
    a.nparts, a.B, a.H, a.W, a.Cout = nparts, b, h, w, cout
    a.clamp = -1.0 if clamp is None else float(clamp)
    L.check(L.lib().hfagp_torgb_finish_fwd(C.byref(a), _stream()), "torgb_finish_fwd")
    return out


def _up_layer_args(x, wt, cout, styles, dcoef, noise, noise_strength, bias, act, alpha, gain, clamp, batch, x_absmax,
                   y_absmax, y_f16):
    x_f16 = x.dtype == torch.float16
    xb, h, w, cin = x.shape
    b = batch if batch is not None else xb
    a = L.ModconvArgs()
    a.x, a.wt = x.data_ptr(), wt.data_ptr()
    if wt.dtype == torch.float16:
        a.precision = PREC_F16 if wt.shape[0] == 1 else PREC_F16X3
    elif wt.dtype == torch.bfloat16:
        a.precision = PREC_BF16X3 if wt.shape[0] == 2 else PREC_BF16X6
    else:
        a.precision = PREC_F32
    a.styles, a.dcoef, a.noise, a.bias = _ptr(styles), _ptr(dcoef), _ptr(noise), _ptr(bias)
    a.x_batch_stride = 0 if (xb == 1 and b > 1) else h * w * cin
    a.B, a.H, a.W, a.Cin, a.Cout = b, h, w, cin, cout
    a.mode, a.act, a.ksplit = CONVT3X3_UP2, _ACT[act], 0
    a.noise_strength, a.alpha, a.gain = noise_strength, alpha, gain
    a.clamp = -1.0 if clamp is None else float(clamp)
    a.x_absmax, a.y_absmax = _ptr(x_absmax), _ptr(y_absmax)
    a.x_f16, a.y_f16 = int(x_f16), int(y_f16)
    return a, b, h, w


_FIR_SCRATCH = {}        # (device, stream) -> scratch tensor of upconv_fir, grown on demand to the next power of two
_FIR_RETIRED = []        # outgrown scratch tensors stay alive: a HIP graph captured earlier may still replay against them
_FIR_BYTES = {}          # (shape, precision, storage, developer switches) -> hfagp_upconv_fir_scratch_bytes (0 = not supported)


def _fir_scratch_bytes(a, x_f16: bool) -> int:
    """hfagp_upconv_fir_scratch_bytes, memoised per layer shape: the library builds the whole launch plan to answer, and the
    generator asks twice per up-sampling layer call (supported? / how much scratch?)."""
    key = (a.B, a.H, a.W, a.Cin, a.Cout, a.precision, x_f16, a.x_batch_stride == 0,
           os.environ.get("HFAGP_DEV_FIR_MIN_BLOCKS"), os.environ.get("HFAGP_DEV_FIR_NSEG"), os.environ.get("HFAGP_DEV_FIR_LEAN"))
    n = _FIR_BYTES.get(key)
    if n is None:
        n = _FIR_BYTES[key] = int(L.lib().hfagp_upconv_fir_scratch_bytes(C.byref(a)))
    return n


def upconv_fir_supported(x: torch.Tensor, wt: torch.Tensor, cout: int, batch: Optional[int] = None) -> bool:
    """Whether `upconv_fir` takes this up-sampling layer (16-bit weight image with one or two parts, Cin % 16 == 0,
    Cout % 128 == 0 and a launch that fills the chip): the library's own rule, hfagp_upconv_fir_scratch_bytes() > 0."""
    if wt.dtype == torch.float32 or not x.is_cuda:
        return False
    if (x.dtype == torch.float16) and not (wt.dtype == torch.float16 and wt.shape[0] == 1):
        return False
    a, *_ = _up_layer_args(x, wt, cout, None, None, None, 0.0, None, "linear", 0.2, 1.0, None, batch, None, None, False)
    return _fir_scratch_bytes(a, x.dtype == torch.float16) > 0


def upconv_fir(x: torch.Tensor, wt: torch.Tensor, cout: int, styles: Optional[torch.Tensor], dcoef: Optional[torch.Tensor],
               noise: Optional[torch.Tensor], noise_strength: float, bias: Optional[torch.Tensor], act: str = "lrelu",
               alpha: float = 0.2, gain: float = math.sqrt(2.0), clamp: Optional[float] = None, batch: Optional[int] = None,
               x_absmax: Optional[torch.Tensor] = None, y_absmax: Optional[torch.Tensor] = None,
               y_f16: bool = False) -> torch.Tensor:
    """The up-sampling layer of a synthesis block in one pass: x [B|1, H, W, Cin] -> [B, 2H, 2W, Cout] =
    bias_act(FIR(conv_transpose2d(x * styles, W, stride 2)) * dcoef + noise) (EG3D conv2d_resample(up=2) + bias_act), the raw
    transposed-conv result never leaving the chip (`modconv(mode=CONVT3X3_UP2)` + `upfir_epilogue` move it through HBM in
    fp32).  Only where `upconv_fir_supported` says so."""
    if x.dtype != torch.float16:
        _chk(x, "x")
    elif not (x.is_cuda and x.is_contiguous()):
        raise RuntimeError("x: expected a contiguous CUDA/ROCm tensor")
    if wt.dtype == torch.float32 or not (wt.is_cuda and wt.is_contiguous() and wt.dim() == 5):
        raise RuntimeError("upconv_fir: needs a 16-bit weight image from weight_prep_prec")
    a, b, h, w = _up_layer_args(x, wt, cout, styles, dcoef, noise, noise_strength, bias, act, alpha, gain, clamp, batch,
                                x_absmax, y_absmax, y_f16)
    nbytes = _fir_scratch_bytes(a, x.dtype == torch.float16)
    if nbytes == 0:
        raise RuntimeError("upconv_fir: this layer shape / precision / batch is not supported (upconv_fir_supported)")
    key = (x.device, torch.cuda.current_stream(x.device).cuda_stream)
    scratch = _FIR_SCRATCH.get(key)
    if scratch is None or scratch.numel() * 4 < nbytes:
        if scratch is not None:
            _FIR_RETIRED.append(scratch)
        grown = 1 << max(nbytes - 1, 1).bit_length()           # powers of two: a handful of growth steps per process at most
        scratch = _FIR_SCRATCH[key] = torch.empty(grown // 4, device=x.device, dtype=torch.float32)
    y = torch.empty(b, 2 * h, 2 * w, cout, device=x.device, dtype=torch.float16 if y_f16 else torch.float32)
    a.y = y.data_ptr()
    L.check(L.lib().hfagp_upconv_fir_fwd(C.byref(a), _ptr(scratch), _stream()), "upconv_fir_fwd")
    return y


def upfir_epilogue(yt: torch.Tensor, dcoef: Optional[torch.Tensor], noise: Optional[torch.Tensor],
                   noise_strength: float, bias: Optional[torch.Tensor], act: str = "lrelu", alpha: float = 0.2,
                   gain: float = math.sqrt(2.0), clamp: Optional[float] = None,
                   y_absmax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """yt [B, 2H+1, 2W+1, C] raw transposed conv → FIR(pad 1, gain 4) → demod/noise/bias/act → [B,2H,2W,C].
    A float16 ``yt`` (fp16 storage, `modconv` y_f16) gives a float16 result; the arithmetic is fp32 either way."""
    half = yt.dtype == torch.float16
    if half:
        if not (yt.is_cuda and yt.is_contiguous()):
            raise RuntimeError("yt: expected a contiguous CUDA/ROCm tensor")
    else:
        _chk(yt, "yt")
    b, hi, wi, c = yt.shape
    h, w = (hi - 1) // 2, (wi - 1) // 2
    y = torch.empty(b, 2 * h, 2 * w, c, device=yt.device, dtype=yt.dtype)
    a = L.UpfirEpilogueArgs()
    a.yt, a.dcoef, a.noise, a.bias, a.y = yt.data_ptr(), _ptr(dcoef), _ptr(noise), _ptr(bias), y.data_ptr()
    a.io_f16 = int(half)
    a.B, a.H, a.W, a.C, a.act = b, h, w, c, _ACT[act]
    a.noise_strength, a.alpha, a.gain = noise_strength, alpha, gain
    a.clamp = -1.0 if clamp is None else float(clamp)
    a.y_absmax = _ptr(y_absmax)
    L.check(L.lib().hfagp_upfir_epilogue_fwd(C.byref(a), _stream()), "upfir_epilogue_fwd")
    return y


def skip_upsample_add(img: Optional[torch.Tensor], y: torch.Tensor, plane_major: bool = False,
                      out_absmax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """SynthesisBlock 'skip': upsample2d(img) + y (channels-last).  plane_major → [B,3,H,W,C/3].  out_absmax ([64 x 32],
    `absmax_slots`): receives max |output| (the bound the ray marcher's 16-bit decoder scales by)."""
    _chk(y, "y")
    b, ho, wo, c = y.shape
    a = L.SkipArgs()
    if img is not None:
        _chk(img, "img")
        a.img_in, a.H, a.W = _ptr(img), img.shape[1], img.shape[2]
    else:
        a.H, a.W = ho, wo
    out = torch.empty((b, 3, ho, wo, c // 3) if plane_major else (b, ho, wo, c), device=y.device,
                      dtype=torch.float32)
    a.y, a.img_out, a.B, a.C, a.plane_major = _ptr(y), _ptr(out), b, c, int(plane_major)
    a.out_absmax = _ptr(out_absmax)
    L.check(L.lib().hfagp_skip_upsample_add(C.byref(a), _stream()), "skip_upsample_add")
    return out


def torgb_skip_supported(x: torch.Tensor, wt: torch.Tensor, cout: int) -> bool:
    """Whether `torgb_skip` (hfagp_torgb_skip_fwd, the streaming toRGB + skip kernel) takes this layer: a 16-bit weight
    image, Cout a multiple of 32 (<= 128), Cin a multiple of 16 (<= 512), rows of a multiple of 32 positions, and enough
    positions to fill the chip."""
    b, h, w, cin = x.shape
    # (below 512 wave tiles of 32 positions the K loop's HBM latency is not covered by other waves — at one frame the
    # 32^2 and 64^2 blocks take 46 / 42 us here against 24 / 27 us for the split-K conv + reducer + skip kernels)
    return (wt.dtype in (torch.float16, torch.bfloat16) and cout % 32 == 0 and cout <= 128 and cin % 16 == 0 and
            cin <= 512 and w % 32 == 0 and (h * w) % 128 == 0 and b * h * w >= 512 * 32)


def torgb_skip(x: torch.Tensor, wt: torch.Tensor, cout: int, styles: torch.Tensor, bias: torch.Tensor,
               img: Optional[torch.Tensor], plane_major: bool = False, x_absmax: Optional[torch.Tensor] = None,
               out_absmax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """SynthesisBlock 'skip' branch in one pass: upsample2d(img) + toRGB(x) (1x1 modulated conv, linear, bias), x and img
    channels-last; returns [B,H,W,Cout] or, plane_major, [B,3,H,W,Cout/3].  ``wt``: `weight_prep_prec` image of the toRGB
    weight.  Same bits as `modconv(CONV1X1)` + `skip_upsample_add` (when that conv is not split along K)."""
    _chk(x, "x")
    b, h, w, cin = x.shape
    if not (wt.is_cuda and wt.is_contiguous() and wt.dim() == 5 and wt.dtype in (torch.float16, torch.bfloat16)):
        raise RuntimeError("torgb_skip: the weight image must come from weight_prep_prec (16-bit kinds)")
    a = L.TorgbSkipArgs()
    if wt.dtype == torch.float16:
        a.precision = PREC_F16 if wt.shape[0] == 1 else PREC_F16X3
    else:
        a.precision = PREC_BF16X3 if wt.shape[0] == 2 else PREC_BF16X6
    out = torch.empty((b, 3, h, w, cout // 3) if plane_major else (b, h, w, cout), device=x.device, dtype=torch.float32)
    a.x, a.wt, a.styles, a.bias = _ptr(x), wt.data_ptr(), _ptr(_chk(styles, "styles")), _ptr(_chk(bias, "bias"))
    if img is not None:
        _chk(img, "img")
        if img.shape != (b, h // 2, w // 2, cout):
            raise RuntimeError(f"torgb_skip: img {tuple(img.shape)} is not the half-resolution image of {(b, h, w, cout)}")
        a.img_in = _ptr(img)
    a.img_out, a.x_absmax, a.out_absmax = _ptr(out), _ptr(x_absmax), _ptr(out_absmax)
    a.B, a.H, a.W, a.Cin, a.Cout, a.plane_major = b, h, w, cin, cout, int(plane_major)
    L.check(L.lib().hfagp_torgb_skip_fwd(C.byref(a), _stream()), "torgb_skip_fwd")
    return out


def torgb_small(x: torch.Tensor, weight: torch.Tensor, styles: torch.Tensor, bias: torch.Tensor,
                rgb_in: Optional[torch.Tensor], clamp: Optional[float], y_pre: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ToRGBLayer with ≤4 output channels + skip add; x channels-last, rgb NCHW.  `y_pre` (optional,
    [B,Cout,H,W]) receives the toRGB output before clamp and skip add (backward needs the clamp mask)."""
    _chk(x, "x")
    b, h, w, cin = x.shape
    cout = weight.shape[0]
    out = torch.empty(b, cout, h, w, device=x.device, dtype=torch.float32)
    a = L.TorgbArgs()
    a.y_pre = _ptr(y_pre)
    a.x, a.weight, a.styles, a.bias = _ptr(x), _ptr(_chk(weight, "weight")), _ptr(_chk(styles, "styles")), _ptr(bias)
    a.rgb_in = _ptr(_chk(rgb_in, "rgb_in")) if rgb_in is not None else None
    a.rgb_out = _ptr(out)
    a.B, a.H, a.W, a.Cin, a.Cout = b, h, w, cin, cout
    a.clamp = -1.0 if clamp is None else float(clamp)
    L.check(L.lib().hfagp_torgb_fwd(C.byref(a), _stream()), "torgb_fwd")
    return out


# ----------------------------------------------------------------------------- renderer
def raymarch(planes: torch.Tensor, cam2world: torch.Tensor, intrinsics: torch.Tensor, u_strat: torch.Tensor,
             u_imp: torch.Tensor, dec_w0: torch.Tensor, dec_b0: torch.Tensor, dec_w1: torch.Tensor,
             dec_b1: torch.Tensor, res: int, ray_start: float, ray_end: float, box_warp: float,
             decoder_lr_mul: float = 1.0, plane_axes: int = 0, white_back: bool = False,
             decoder_precision: str = "f16x3", planes_absmax: Optional[torch.Tensor] = None,
             state: Optional[torch.Tensor] = None):
    """planes [B,3,H,W,32] → feat [B,R,32], depth [B,R] (unclamped), wsum [B,R], tminmax [B,R,2].
    ``state`` (`raymarch_state`): receives the per-sample colours / densities / depths / sort order of every ray, so that
    `raymarch_bwd` of the same step need not gather and decode every sample again for the compositing adjoint.
    decoder_precision 'f16x3': the decoder MLP on the 16-bit matrix pipe with split fp16 operands (fp32-class); it needs
    a bound on |planes| — `planes_absmax` (64 slots as published by `skip_upsample_add(out_absmax=...)`), computed here
    with one reduction over the planes when the caller has none.  'fp32': the exact fp32 matrix instructions."""
    _chk(planes, "planes")
    b, three, h, w, ch = planes.shape
    if three != 3 or ch != 32:
        raise RuntimeError("raymarch: planes must be [B, 3, H, W, 32]")
    r = res * res
    sc, sf = u_strat.shape[-1], u_imp.shape[-1]
    if u_strat.numel() != b * r * sc or u_imp.numel() != b * r * sf:
        raise RuntimeError("raymarch: u_strat / u_imp have the wrong number of elements")
    dev = planes.device
    feat = torch.empty(b, r, 32, device=dev, dtype=torch.float32)
    depth = torch.empty(b, r, device=dev, dtype=torch.float32)
    wsum = torch.empty(b, r, device=dev, dtype=torch.float32)
    tmm = torch.empty(b, r, 2, device=dev, dtype=torch.float32)
    a = L.RaymarchArgs()
    a.planes, a.cam2world, a.intrinsics = _ptr(planes), _ptr(_chk(cam2world, "cam2world")), _ptr(_chk(intrinsics, "intrinsics"))
    a.u_strat, a.u_imp = _ptr(_chk(u_strat, "u_strat")), _ptr(_chk(u_imp, "u_imp"))
    a.dec_w0, a.dec_b0 = _ptr(_chk(dec_w0, "dec_w0")), _ptr(_chk(dec_b0, "dec_b0"))
    a.dec_w1, a.dec_b1 = _ptr(_chk(dec_w1, "dec_w1")), _ptr(_chk(dec_b1, "dec_b1"))
    a.feat, a.depth, a.wsum, a.tminmax = _ptr(feat), _ptr(depth), _ptr(wsum), _ptr(tmm)
    a.B, a.H, a.W, a.res, a.Sc, a.Sf = b, h, w, res, sc, sf
    a.plane_axes, a.white_back = plane_axes, int(white_back)
    a.ray_start, a.ray_end, a.box_warp, a.decoder_lr_mul = ray_start, ray_end, box_warp, decoder_lr_mul
    a.planes_absmax = _ptr(_decoder_bound(planes, decoder_precision, planes_absmax))
    if state is not None:
        if state.shape != (b, r, (sc + sf) * 35):
            raise RuntimeError(f"raymarch: state must be [B, R, {(sc + sf) * 35}] (raymarch_state), got {tuple(state.shape)}")
        a.state = _ptr(_chk(state, "state"))
    L.check(L.lib().hfagp_raymarch_fwd(C.byref(a), _stream()), "raymarch_fwd")
    return feat, depth, wsum, tmm


def depth_clamp_(depth: torch.Tensor, tminmax: torch.Tensor) -> torch.Tensor:
    """In place: depth.clamp_(tminmax[..., 0].min(), tminmax[..., 1].max()) — MipRayMarcher2's batch-global depth clamp — as ONE
    launch (hfagp_depth_clamp)."""
    _chk(depth, "depth")
    _chk(tminmax, "tminmax")
    n = depth.numel()
    L.check(L.lib().hfagp_depth_clamp(_ptr(depth), _ptr(tminmax), n, _stream()), "depth_clamp")
    return depth


def raymarch_state(b: int, res: int, sc: int, sf: int, device) -> torch.Tensor:
    """Buffer for `raymarch(..., state=)` / `raymarch_bwd(..., state=)`: [B, R, 35 (Sc + Sf)] floats = 13.4 KB per ray at 48 + 48."""
    return torch.empty(b, res * res, (sc + sf) * 35, device=device, dtype=torch.float32)


def _decoder_bound(planes: torch.Tensor, decoder_precision: str, planes_absmax: Optional[torch.Tensor]):
    """The 64-slot bound on |planes| of the 16-bit decoder (None for the exact fp32 decoder)."""
    if decoder_precision == "fp32":
        return None
    if decoder_precision != "f16x3":
        raise ValueError(f"decoder_precision must be 'fp32' or 'f16x3', got {decoder_precision!r}")
    if planes_absmax is None:
        planes_absmax = planes.abs().amax().expand(L.ABSMAX_FLOATS).contiguous()
    return _chk(planes_absmax, "planes_absmax")


# ----------------------------------------------------------------------------- standalone ops (NCHW)
def _upfirdn2d_raw(x, f, up, down, padding, gain):
    n, c, h, w = x.shape
    fh, fw = f.shape
    px0, px1, py0, py1 = padding
    ho = (h * up + py0 + py1 - fh) // down + 1
    wo = (w * up + px0 + px1 - fw) // down + 1
    y = torch.empty(n, c, ho, wo, device=x.device, dtype=torch.float32)
    L.check(L.lib().hfagp_upfirdn2d_fwd(_ptr(x), _ptr(f), _ptr(y), n, c, h, w, fh, fw, up, down,
                                        px0, px1, py0, py1, gain, _stream()), "upfirdn2d_fwd")
    return y


class _UpFirDn2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f, up, down, padding, gain):
        ctx.save_for_backward(f)
        ctx.cfg = (x.shape, up, down, tuple(padding), gain)
        return _upfirdn2d_raw(x, f, up, down, padding, gain)

    @staticmethod
    def backward(ctx, dy):
        (f,) = ctx.saved_tensors
        (n, c, h, w), up, down, (px0, px1, py0, py1), gain = ctx.cfg
        dy = _chk(dy.contiguous(), "dy")
        dx = torch.empty(n, c, h, w, device=dy.device, dtype=torch.float32)
        L.check(L.lib().hfagp_upfirdn2d_bwd(_ptr(dy), _ptr(f), _ptr(dx), n, c, h, w, f.shape[0], f.shape[1], up, down,
                                            px0, px1, py0, py1, gain, _stream()), "upfirdn2d_bwd")
        return dx, None, None, None, None, None


def upfirdn2d(x: torch.Tensor, f: torch.Tensor, up: int = 1, down: int = 1,
              padding=(0, 0, 0, 0), gain: float = 1.0) -> torch.Tensor:
    """EG3D upfirdn2d(x, f, up, down, padding=[px0,px1,py0,py1], gain) on NCHW fp32; differentiable w.r.t. x."""
    _chk(x, "x")
    _chk(f, "f")
    if x.requires_grad and torch.is_grad_enabled():
        return _UpFirDn2d.apply(x, f, up, down, tuple(padding), gain)
    return _upfirdn2d_raw(x, f, up, down, padding, gain)


def upsample2d(x: torch.Tensor, f: torch.Tensor) -> torch.Tensor:
    return upfirdn2d(x, f, up=2, padding=(2, 1, 2, 1), gain=4.0)


def _bias_act_raw(x, b, dim, act, alpha, gain, clamp):
    y = torch.empty_like(x)
    inner = 1
    for s in x.shape[dim + 1:]:
        inner *= s
    L.check(L.lib().hfagp_bias_act_fwd(_ptr(x), _ptr(b), _ptr(y), x.numel(), x.shape[dim], inner, _ACT[act],
                                       alpha, gain, -1.0 if clamp is None else float(clamp), _stream()), "bias_act_fwd")
    return y


class _BiasAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b, dim, act, alpha, gain, clamp):
        y = _bias_act_raw(x, b, dim, act, alpha, gain, clamp)
        ctx.save_for_backward(y)
        ctx.cfg = (dim, act, alpha, gain, clamp, b is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dim, act, alpha, gain, clamp, has_b = ctx.cfg
        dy = _chk(dy.contiguous(), "dy")
        dx = torch.empty_like(y)
        L.check(L.lib().hfagp_bias_act_bwd(_ptr(dy), _ptr(y), _ptr(dx), y.numel(), _ACT[act], alpha, gain,
                                           -1.0 if clamp is None else float(clamp), _stream()), "bias_act_bwd")
        db = dx.sum([d for d in range(dx.dim()) if d != dim]) if has_b else None
        return dx, db, None, None, None, None, None


def bias_act_bwd(dy: torch.Tensor, y: torch.Tensor, act: str, alpha: float, gain: float,
                 clamp: Optional[float]) -> torch.Tensor:
    """Gradient of bias_act w.r.t. its pre-activation input from the OUTPUT y (EG3D's bias_act backward): dy * gain *
    lrelu'(y) * [|y| < clamp]; any layout (elementwise)."""
    _chk(dy, "dy"), _chk(y, "y")
    dx = torch.empty_like(y)
    L.check(L.lib().hfagp_bias_act_bwd(_ptr(dy), _ptr(y), _ptr(dx), y.numel(), _ACT[act], alpha, gain,
                                       -1.0 if clamp is None else float(clamp), _stream()), "bias_act_bwd")
    return dx


def bias_act(x: torch.Tensor, b: Optional[torch.Tensor] = None, dim: int = 1, act: str = "linear",
             alpha: float = 0.2, gain: Optional[float] = None, clamp: Optional[float] = None) -> torch.Tensor:
    """EG3D bias_act.bias_act(x, b, dim, act, alpha, gain, clamp); differentiable w.r.t. x and b."""
    _chk(x, "x")
    if gain is None:
        gain = math.sqrt(2.0) if act == "lrelu" else 1.0
    if torch.is_grad_enabled() and (x.requires_grad or (b is not None and b.requires_grad)):
        return _BiasAct.apply(x, b, dim, act, alpha, gain, clamp)
    return _bias_act_raw(x, b, dim, act, alpha, gain, clamp)


class _BlurDown(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        b, h, w, c = x.shape
        ctx.shape = (b, h, w, c)
        y = torch.empty(b, h // 2, w // 2, c, device=x.device, dtype=torch.float32)
        L.check(L.lib().hfagp_blur_down_fwd(_ptr(x), _ptr(y), b, h, w, c, _stream()), "blur_down_fwd")
        return y

    @staticmethod
    def backward(ctx, gy):
        b, h, w, c = ctx.shape
        gy = _chk(gy.contiguous(), "gy")
        gx = torch.empty(b, h, w, c, device=gy.device, dtype=torch.float32)
        L.check(L.lib().hfagp_blur_down_bwd(_ptr(gy), _ptr(gx), b, h, w, c, _stream()), "blur_down_bwd")
        return gx


def blur_down(x: torch.Tensor) -> torch.Tensor:
    """Blur(pad (1,1), FIR [1,3,3,1]) + stride-2 sampling of a channels-last tensor [B,H,W,C] -> [B,H/2,W/2,C]
    (the front of the RGB driver's 1x1 skip conv, encoder3d.py ConvLayer(downsample=True)); differentiable."""
    _chk(x, "x")
    return _BlurDown.apply(x)


def nchw_to_nhwc(x: torch.Tensor) -> torch.Tensor:
    _chk(x, "x")
    b, c, h, w = x.shape
    y = torch.empty(b, h, w, c, device=x.device, dtype=torch.float32)
    L.check(L.lib().hfagp_nchw_to_nhwc(_ptr(x), _ptr(y), b, c, h, w, _stream()), "nchw_to_nhwc")
    return y


def nhwc_to_nchw(x: torch.Tensor) -> torch.Tensor:
    _chk(x, "x")
    b, h, w, c = x.shape
    y = torch.empty(b, c, h, w, device=x.device, dtype=torch.float32)
    L.check(L.lib().hfagp_nhwc_to_nchw(_ptr(x), _ptr(y), b, c, h, w, _stream()), "nhwc_to_nchw")
    return y


# ----------------------------------------------------------------------------- backward pass
_DEV_PW_CHUNKS = os.environ.get("HFAGP_DEV_PW_CHUNKS")


def pointwise_bwd(x: torch.Tensor, dxs_conv=None, s_conv=None, dxs_rgb=None, s_rgb=None, g_rgb_small=None,
                  w_rgb_small=None, s_small=None, g_direct=None, producer: Optional[dict] = None,
                  param_grads: bool = False, y_rgb_small=None, clamp_rgb_small: Optional[float] = None,
                  g_nchw3_a=None, g_nchw3_b=None, deferred: Optional[list] = None):
    """Fused streaming pass over the saved activation x [B,H,W,C] (see include/hfagp.h).  `producer` =
    dict(dcoef, bias, noise, noise_strength, act, alpha, gain, clamp) of the layer that produced x, or None.
    Returns (g_out [B,H,W,C], sums [B,10,C])."""
    _chk(x, "x")
    b, h, w, c = x.shape
    # blocks per sample: at most 4 pixels per thread (the low-resolution layers are latency-bound otherwise: one block
    # walking 64 pixels of 512 channels took 45 us), at most 256 (the partial sums are reduced by a second kernel)
    npl = max(1, 256 // (c // 4))
    # ... and enough blocks to fill the chip: a workgroup waits out one memory round trip per two pixels of its threads, so what
    # counts is how many are RESIDENT — registers allow 5 / 3 / 4 / 3 per CU for the (small toRGB, parameter gradients) variants
    # (the LDS scratch is sized by the rows a variant reduces since round 5; ten rows for all = 40 KB capped it at 4, and 512
    # blocks in total left two per CU: the 512^2 x 64 launches of the fitting step ran at 1.4 - 2.7 TB/s)
    per_cu = {(False, False): 5, (True, False): 3, (False, True): 4, (True, True): 3}[(g_rgb_small is not None, bool(param_grads))]
    c4 = c // 4
    if g_rgb_small is not None and not param_grads and c4 >= 8 and c4 & (c4 - 1) == 0:
        per_cu = 4                     # (the small-toRGB operands shared by shuffles: 128 registers)
    nchunks = max(1, min(max(256, per_cu * 256 // b), -(-(h * w) // (npl * 4))))
    if _DEV_PW_CHUNKS:                                  # developer sweep (tools/dev/pointwise_sweep.py)
        nchunks = max(1, min(int(_DEV_PW_CHUNKS), h * w))
    a = L.PointwiseBwdArgs()
    g_out = torch.empty_like(x)
    partial = torch.empty(b, nchunks, 10, c, device=x.device, dtype=torch.float32)
    sums = torch.empty(b, 10, c, device=x.device, dtype=torch.float32)
    # `deferred` (a list): the partial sums stay un-reduced and (partial, sums) is appended — `reduce_partials_batch(deferred)`
    # fills every `sums` of the list in one launch (its views may be handed around before that)
    a.x, a.g_out, a.partial, a.sums = _ptr(x), _ptr(g_out), _ptr(partial), (None if deferred is not None else _ptr(sums))
    if deferred is not None:
        deferred.append((partial, sums))
    a.dxs_conv, a.s_conv, a.dxs_rgb, a.s_rgb = _ptr(dxs_conv), _ptr(s_conv), _ptr(dxs_rgb), _ptr(s_rgb)
    a.g_rgb_small, a.w_rgb_small, a.s_small, a.g_direct = _ptr(g_rgb_small), _ptr(w_rgb_small), _ptr(s_small), _ptr(g_direct)
    a.B, a.H, a.W, a.C, a.nchunks = b, h, w, c, nchunks
    a.Co = g_rgb_small.shape[1] if g_rgb_small is not None else 0
    a.y_rgb_small = _ptr(y_rgb_small) if (y_rgb_small is not None and clamp_rgb_small is not None) else None
    a.clamp_rgb_small = -1.0 if clamp_rgb_small is None else float(clamp_rgb_small)
    for t in (g_nchw3_a, g_nchw3_b):
        if t is not None and tuple(t.shape) != (b, 3, h, w):
            raise RuntimeError(f"pointwise_bwd: g_nchw3_* must be [B, 3, H, W] = {(b, 3, h, w)}, got {tuple(t.shape)}")
    a.g_nchw3_a, a.g_nchw3_b = _ptr(g_nchw3_a), _ptr(g_nchw3_b)
    a.clamp = -1.0
    a.param_grads = int(param_grads)
    if producer is not None:
        a.has_producer = 1
        a.dcoef_p, a.bias_p, a.noise_p = _ptr(producer.get("dcoef")), _ptr(producer.get("bias")), _ptr(producer.get("noise"))
        a.noise_strength_p = producer.get("noise_strength", 0.0)
        a.noise_strength_dev = _ptr(producer.get("noise_strength_dev"))     # (0-d device tensor: wins over the host value)
        a.act_p = _ACT[producer.get("act", "lrelu")]
        a.alpha, a.gain = producer.get("alpha", 0.2), producer.get("gain", math.sqrt(2.0))
        clamp = producer.get("clamp")
        a.clamp = -1.0 if clamp is None else float(clamp)
    L.check(L.lib().hfagp_pointwise_bwd(C.byref(a), _stream()), "pointwise_bwd")
    return g_out, sums


def upfir_bwd(g_y: torch.Tensor) -> torch.Tensor:
    """g_y [B,2H,2W,C] → parity images of the y_t gradient [2,2,B,H+1,W+1,C]."""
    _chk(g_y, "g_y")
    b, ho, wo, c = g_y.shape
    h, w = ho // 2, wo // 2
    gph = torch.empty(2, 2, b, h + 1, w + 1, c, device=g_y.device, dtype=torch.float32)
    L.check(L.lib().hfagp_upfir_bwd(_ptr(g_y), _ptr(gph), b, h, w, c, _stream()), "upfir_bwd")
    return gph


def upsample2d_bwd(g: torch.Tensor, channels_last: bool) -> torch.Tensor:
    """Adjoint of upsample2d: [B,2H,2W,C] → [B,H,W,C] (channels_last) or [B,C,2H,2W] → [B,C,H,W]."""
    _chk(g, "g")
    if channels_last:
        b, ho, wo, c = g.shape
        out = torch.empty(b, ho // 2, wo // 2, c, device=g.device, dtype=torch.float32)
        outer, inner = b, c
    else:
        b, c, ho, wo = g.shape
        out = torch.empty(b, c, ho // 2, wo // 2, device=g.device, dtype=torch.float32)
        outer, inner = b * c, 1
    L.check(L.lib().hfagp_upsample2d_bwd(_ptr(g), _ptr(out), outer, ho // 2, wo // 2, inner, _stream()), "upsample2d_bwd")
    return out


def planes_to_nhwc(pm: torch.Tensor) -> torch.Tensor:
    _chk(pm, "planes")
    b, three, h, w, cp = pm.shape
    y = torch.empty(b, h, w, 3 * cp, device=pm.device, dtype=torch.float32)
    L.check(L.lib().hfagp_planes_to_nhwc(_ptr(pm), _ptr(y), b, h, w, cp, _stream()), "planes_to_nhwc")
    return y


def style_bwd(ds: torch.Tensor, dd: Optional[torch.Tensor], styles: torch.Tensor, dcoef: Optional[torch.Tensor],
              wsq: Optional[torch.Tensor], affine_w: torch.Tensor, dw: torch.Tensor, style_gain: float = 1.0,
              accumulate: bool = True) -> torch.Tensor:
    """Accumulate d ws for one layer into the row view dw [B, w_dim] (strided view of d_ws); returns
    dstot [B, Cin] = gradient w.r.t. the raw affine output (input of affine_grad)."""
    b, cin = styles.shape
    a = L.StyleBwdArgs()
    dstot = torch.empty(b, cin, device=styles.device, dtype=torch.float32)
    if not ds.is_contiguous():
        ds = ds.contiguous()
    if dd is not None and not dd.is_contiguous():
        dd = dd.contiguous()
    a.ds, a.dd, a.styles, a.dcoef, a.wsq = _ptr(ds), _ptr(dd), _ptr(styles), _ptr(dcoef), _ptr(wsq)
    a.affine_w, a.dstot, a.dw = _ptr(affine_w), _ptr(dstot), dw.data_ptr()
    a.B, a.Cin, a.Cout = b, cin, (dd.shape[1] if dd is not None else 0)
    a.w_dim, a.dw_stride, a.accumulate = affine_w.shape[1], dw.stride(0), int(accumulate)
    a.style_gain = style_gain
    L.check(L.lib().hfagp_style_bwd(C.byref(a), _stream()), "style_bwd")
    return dstot


def style_bwd_batch(items, d_ws: torch.Tensor):
    """All style gradients of a backward pass in two launches.  ``items``: list of (ds [B,Cin] view, dd [B,Cout] view |
    None, styles, dcoef | None, wsq | None, affine_w, row, style_gain); ds / dd may be strided row views (unit inner
    stride) of the reduction tensors.  Accumulates into d_ws[:, row]; returns the dstot [B, Cin] of every item, in the
    order given."""
    order = sorted(range(len(items)), key=lambda i: items[i][6])          # by ws row = by dw pointer
    out = [None] * len(items)
    for c0 in range(0, len(order), 32):
        chunk = order[c0:c0 + 32]
        # a ws row must not straddle two launches (its accumulation is one block's job): cut the chunk at a row boundary
        while c0 + len(chunk) < len(order) and len(chunk) > 1 and items[chunk[-1]][6] == items[order[c0 + len(chunk)]][6]:
            chunk = chunk[:-1]
        if len(chunk) != len(order[c0:c0 + 32]):
            return _style_bwd_each(items, d_ws)                            # (never with <= 32 layers: ffhq has 26)
        arr = (L.StyleBwdItem * len(chunk))()
        for j, i in enumerate(chunk):
            ds, dd, styles, dcoef, wsq, affine_w, row, gain = items[i]
            if ds.stride(-1) != 1 or (dd is not None and dd.stride(-1) != 1):
                raise RuntimeError("style_bwd_batch: ds / dd need unit inner stride")
            b, cin = styles.shape
            dstot = torch.empty(b, cin, device=styles.device, dtype=torch.float32)
            a = arr[j]
            a.ds, a.dd, a.styles, a.dcoef, a.wsq = ds.data_ptr(), (dd.data_ptr() if dd is not None else None), _ptr(styles), \
                _ptr(dcoef), _ptr(wsq)
            a.affine_w, a.dstot = _ptr(affine_w), _ptr(dstot)
            a.dw = d_ws.data_ptr() + row * d_ws.stride(1) * 4
            a.B, a.Cin, a.Cout = b, cin, (dd.shape[1] if dd is not None else 0)
            a.w_dim, a.dw_stride = affine_w.shape[1], d_ws.stride(0)
            a.ds_stride, a.dd_stride = ds.stride(0), (dd.stride(0) if dd is not None else 0)
            a.style_gain = gain
            out[i] = dstot
        L.check(L.lib().hfagp_style_batch_bwd(arr, len(chunk), _stream()), "style_batch_bwd")
    return out


def _style_bwd_each(items, d_ws):
    return [style_bwd(ds, dd, styles, dcoef, wsq, affine_w, d_ws[:, row], gain, accumulate=True)
            for ds, dd, styles, dcoef, wsq, affine_w, row, gain in items]


def conv_wgrad(x: torch.Tensor, styles: Optional[torch.Tensor], g: torch.Tensor, weight: torch.Tensor, mode: int,
               dd: Optional[torch.Tensor] = None, dcoef: Optional[torch.Tensor] = None,
               precision: str = "fp32", ksplit: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Weight gradient of a modulated conv (see include/hfagp.h): returns dweight like `weight`.  precision 'bf16x3':
    the 3x3 and up-sampling modes with Cin, Cout multiples of 64 run on the split-bf16 MFMA kernel (the 1x1 mode and
    other shapes stay fp32).  `out`: ACCUMULATE into this tensor (the parameter's .grad slice of the trainer's flat buffer)
    instead of returning a new one."""
    _chk(x, "x")
    _chk(g, "g")
    b, h, w, cin = x.shape
    cout = weight.shape[0]
    a = L.WgradArgs()
    if out is not None:
        if out.shape != weight.shape or out.dtype != torch.float32 or not out.is_contiguous():
            raise RuntimeError("conv_wgrad: `out` must be a contiguous fp32 tensor of the weight's shape")
        if out.data_ptr() % 16 != 0:
            # (the reducers read and write dW with 16-byte accesses; FlatGrads aligns every slice, a caller's own view
            # may not be)
            raise RuntimeError("conv_wgrad: `out` must be 16-byte aligned (trainer.FlatGrads slices are)")
        a.accumulate = 1
    dweight = torch.empty_like(weight) if out is None else out
    a.x, a.styles, a.g = _ptr(x), _ptr(styles), _ptr(g)
    a.weight, a.dd, a.dcoef, a.dweight = _ptr(_chk(weight.detach(), "weight")), _ptr(dd), _ptr(dcoef), _ptr(dweight)
    if dd is not None:                  # a row view [B, Cout] of pointwise_bwd's sums is read in place (ABI 12 `dd_stride`)
        if dd.dim() != 2 or dd.shape[1] != cout or dd.stride(1) != 1 or dd.dtype != torch.float32:
            raise RuntimeError("conv_wgrad: dd must be [B, Cout] fp32 with unit column stride")
        a.dd_stride = dd.stride(0)
    a.B, a.H, a.W, a.Cin, a.Cout, a.mode = b, h, w, cin, cout, mode
    a.precision = PREC_BF16X3 if precision == "bf16x3" else PREC_F32
    a.ksplit = L.lib().hfagp_wgrad_ksplit(C.byref(a)) if ksplit is None else ksplit      # (the library's split-K policy)
    ws = torch.empty(L.lib().hfagp_wgrad_workspace_bytes(C.byref(a)) // 4, device=x.device, dtype=torch.float32)
    a.workspace = _ptr(ws)
    L.check(L.lib().hfagp_conv_wgrad(C.byref(a), _stream()), "conv_wgrad")
    return dweight


def affine_grad(dstot: torch.Tensor, w: torch.Tensor, dA: torch.Tensor, db: torch.Tensor) -> None:
    """dA [Cin, w_dim] += dstot^T . w / sqrt(w_dim); db [Cin] += sum_b dstot.  w is a row view [B, w_dim] of ws."""
    b, cin = dstot.shape
    L.check(L.lib().hfagp_affine_grad(_ptr(dstot), w.data_ptr(), _ptr(dA), _ptr(db), b, cin, w.shape[1], w.stride(0),
                                      _stream()), "affine_grad")


def reduce_partials_batch(items) -> None:
    """items: (partial [B,nchunks,10,C], sums [B,10,C]) pairs of `pointwise_bwd(..., deferred=list)` calls: one launch per 32."""
    for i0 in range(0, len(items), 32):
        chunk = items[i0:i0 + 32]
        arr = (L.ReducePartialsItem * len(chunk))()
        for a, (partial, sums) in zip(arr, chunk):
            a.partial, a.sums = _ptr(_chk(partial, "partial")), _ptr(_chk(sums, "sums"))
            a.B, a.nchunks, a.n = partial.shape[0], partial.shape[1], partial.shape[2] * partial.shape[3]
        L.check(L.lib().hfagp_reduce_partials_batch(arr, len(chunk), _stream()), "reduce_partials_batch")
    items.clear()


def affine_grad_batch(items) -> None:
    """items: (dstot [B,Cin], w row view [B,w_dim] of ws, dA [Cin,w_dim], db [Cin]) per affine layer; dA / db are accumulated
    into (the parameters' .grad slices): every affine layer of a backward pass in one launch per 32."""
    for i0 in range(0, len(items), 32):
        chunk = items[i0:i0 + 32]
        arr = (L.AffineGradItem * len(chunk))()
        for a, (dstot, w, dA, db) in zip(arr, chunk):
            a.dstot, a.w, a.dA, a.db = _ptr(_chk(dstot, "dstot")), w.data_ptr(), _ptr(_chk(dA, "dA")), _ptr(_chk(db, "db"))
            a.B, a.Cin, a.w_dim, a.w_stride = dstot.shape[0], dstot.shape[1], w.shape[1], w.stride(0)
        L.check(L.lib().hfagp_affine_grad_batch(arr, len(chunk), _stream()), "affine_grad_batch")


def bias_noise_grads(items) -> None:
    """items: (sums [B,10,C] of pointwise_bwd(param_grads=True), dbias [C] or None, dnoise [1] or None): accumulated into."""
    for i0 in range(0, len(items), 32):
        chunk = items[i0:i0 + 32]
        arr = (L.BiasNoiseGradItem * len(chunk))()
        for a, (sums, dbias, dnoise) in zip(arr, chunk):
            a.sums, a.dbias, a.dnoise = _ptr(_chk(sums, "sums")), _ptr(dbias), _ptr(dnoise)
            a.B, a.C = sums.shape[0], sums.shape[2]
        L.check(L.lib().hfagp_bias_noise_grads(arr, len(chunk), _stream()), "bias_noise_grads")


def channel_sum(g: torch.Tensor, out: torch.Tensor, accumulate: bool = True) -> None:
    """out[c] (+)= sum over all leading dims of the channels-last tensor g [..., C]."""
    _chk(g, "g")
    c = g.shape[-1]
    npix = g.numel() // c
    # the kernel walks the tensor as a flat array with a grid stride that is a multiple of C (a thread keeps one channel):
    # nblocks = a multiple of C / gcd(C, 256), about one block per 4096 floats, at most ~1024
    unit = c // math.gcd(c, 256)
    nblocks = unit * int(max(1, min(1024 // unit, g.numel() // (4096 * unit))))
    partial = torch.empty(nblocks, c, device=g.device, dtype=torch.float32)
    L.check(L.lib().hfagp_channel_sum(_ptr(g), _ptr(partial), _ptr(out), npix, c, nblocks, int(accumulate), _stream()),
            "channel_sum")


ROWS_STATS = {"bytes": 0, "chunks": 0, "path": None}      # of the last raymarch_bwd call (bench.py reports it)
_WARNED = set()


def _warn_once(key: str, msg: str) -> None:
    if key not in _WARNED:
        _WARNED.add(key)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def rows_scratch_cap(device=None) -> int:
    """Bytes the sort + gather ray-march backward may take for its scratch in ONE call: HFAGP_RAYBWD_SCRATCH_GIB (default 16), and
    never more than half of what the device has free right now (plus what torch's allocator already caches)."""
    cap = int(float(os.environ.get("HFAGP_RAYBWD_SCRATCH_GIB", "16")) * (1 << 30))
    try:
        free, _ = torch.cuda.mem_get_info(device)
        free += torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        cap = min(cap, max(free // 2, 1 << 28))
    except Exception:  # noqa: BLE001 — no device query: the configured cap alone
        pass
    return cap


def raymarch_bwd(g_feat: torch.Tensor, planes: torch.Tensor, cam2world, intrinsics, u_strat, u_imp, dec_w0, dec_b0,
                 dec_w1, dec_b1, res: int, ray_start: float, ray_end: float, box_warp: float,
                 decoder_lr_mul: float = 1.0, plane_axes: int = 0, white_back: bool = False,
                 return_rec: bool = False, decoder_grads: bool = False, decoder_precision: str = "f16x3",
                 planes_absmax: Optional[torch.Tensor] = None, state: Optional[torch.Tensor] = None,
                 two_kernel: bool = False, rows: Optional[bool] = None, dec_out=None, _out=None):
    """g_feat [B,R,32] → d planes [B,3,H,W,32].  ``state``: what the forward call of this step left behind
    (`raymarch(..., state=)`): the compositing adjoint reads it instead of recomputing.  ``rows``: pass 2 as sort + gather
    (hfagp.h `rows_scratch`, csrc/raymarch_rows.hip) — None = whenever it applies (HFAGP_RAYBWD_ROWS=0 turns the default off: A/B
    timing), False = the scatter kernels.  The sort's scratch (~0.9 GB per frame at 128^2 rays x 96 samples) is bounded by
    `rows_scratch_cap()`: a batch that needs more is processed in frame chunks (round 6; rounds 5 fell back to the 2 x slower scatter
    kernels above 16 GiB, i.e. from B = 19 on, without a word); where the sort does not apply at all (more than 8192 bins per frame)
    or its scratch cannot be allocated, the scatter kernels run and say so once."""
    _chk(planes, "planes")
    _chk(g_feat, "g_feat")
    b, _, h, w, _ = planes.shape
    r = res * res
    sc, sf = u_strat.shape[-1], u_imp.shape[-1]
    if _out is not None:
        d_planes, rec = _out
    else:
        d_planes = torch.zeros_like(planes)
        rec = torch.empty(b, r, sc + sf, 4, device=planes.device, dtype=torch.float32)
    planes_absmax = _decoder_bound(planes, decoder_precision, planes_absmax)      # (one bound for every chunk of the batch)
    a = L.RaymarchBwdArgs()
    f = a.fwd
    f.planes, f.cam2world, f.intrinsics = _ptr(planes), _ptr(_chk(cam2world, "cam2world")), _ptr(_chk(intrinsics, "intrinsics"))
    f.u_strat, f.u_imp = _ptr(_chk(u_strat, "u_strat")), _ptr(_chk(u_imp, "u_imp"))
    f.dec_w0, f.dec_b0, f.dec_w1, f.dec_b1 = _ptr(dec_w0), _ptr(dec_b0), _ptr(dec_w1), _ptr(dec_b1)
    f.B, f.H, f.W, f.res, f.Sc, f.Sf = b, h, w, res, sc, sf
    f.plane_axes, f.white_back = plane_axes, int(white_back)
    f.ray_start, f.ray_end, f.box_warp, f.decoder_lr_mul = ray_start, ray_end, box_warp, decoder_lr_mul
    f.planes_absmax = _ptr(planes_absmax)
    if state is not None:
        if state.shape != (b, r, (sc + sf) * 35):
            raise RuntimeError(f"raymarch_bwd: state must be [B, R, {(sc + sf) * 35}], got {tuple(state.shape)}")
        f.state = _ptr(_chk(state, "state"))
    a.g_feat, a.d_planes, a.rec = _ptr(g_feat), _ptr(d_planes), _ptr(rec)
    # pass 2 as two kernels (dL/dF through a scratch buffer, then the scatter alone: hfagp.h `df_scratch`) where the column
    # variant applies; 201 MB per frame at 128^2 rays x 96 samples.  OFF by default: measured SLOWER than the fused kernel
    # (B = 2: dL/dF 0.51 ms + scatter 1.41 ms against 1.58 ms fused — the scatter alone is the bound, the arithmetic already
    # hides under it; profiles/r04_raybwd_split.txt).  Kept as the vehicle for work on the scatter.
    if rows is None:
        rows = os.environ.get("HFAGP_RAYBWD_ROWS", "1") != "0" and not two_kernel
    scratch = None
    if rows:
        need = int(L.lib().hfagp_raymarch_bwd_rows_bytes(C.byref(f)))
        # (the free-memory query is a driver call: only batches that could matter pay it — a fitting step's 1.8 GB does not)
        cap = rows_scratch_cap(planes.device) if need > (4 << 30) or "HFAGP_RAYBWD_SCRATCH_GIB" in os.environ else (16 << 30)
        if need > cap and b > 1:
            # frame chunks: every per-frame input / output is a contiguous slice of the batch; the decoder gradients accumulate
            nb = max(1, min(b - 1, int(b * cap // need)))
            if decoder_grads and dec_out is None:
                dec_out = tuple(torch.zeros_like(t) for t in (dec_w0, dec_b0, dec_w1, dec_b1))
            for b0 in range(0, b, nb):
                b1 = min(b, b0 + nb)
                raymarch_bwd(g_feat[b0:b1], planes[b0:b1], cam2world[b0:b1], intrinsics[b0:b1], u_strat[b0:b1], u_imp[b0 * r:b1 * r],
                             dec_w0, dec_b0, dec_w1, dec_b1, res, ray_start, ray_end, box_warp, decoder_lr_mul, plane_axes, white_back,
                             False, decoder_grads, decoder_precision, planes_absmax, None if state is None else state[b0:b1],
                             two_kernel, rows, dec_out, _out=(d_planes[b0:b1], rec[b0:b1]))
            ROWS_STATS.update(bytes=min(need, cap), chunks=-(-b // nb), path="rows")
            if decoder_grads:
                return d_planes, tuple(dec_out)
            return (d_planes, rec) if return_rec else d_planes
        if need > 0:
            try:
                scratch = torch.empty(need, device=planes.device, dtype=torch.uint8)
                a.rows_scratch, a.rows_scratch_bytes = _ptr(scratch), need
                ROWS_STATS.update(bytes=need, chunks=1, path="rows")
            except torch.cuda.OutOfMemoryError:
                _warn_once("rows_oom", f"raymarch_bwd: {need / 2**30:.2f} GiB of sort scratch could not be allocated: this call runs the "
                                       f"(~2 x slower) scatter kernels")
        else:
            _warn_once("rows_unsupported", f"raymarch_bwd: the sort + gather backward does not take {h} x {w} planes at {res}^2 rays "
                                           f"(more than 8192 bins per frame): the (~2 x slower) scatter kernels run")
        if scratch is None:
            ROWS_STATS.update(bytes=0, chunks=1, path="scatter")
    if scratch is None and two_kernel and plane_axes == 0 and h == w and h <= 256 and b * r * (sc + sf) * 128 <= (4 << 30):
        df = torch.empty(b, r, sc + sf, 32, device=planes.device, dtype=torch.float32)
        a.df_scratch = _ptr(df)
    dec = None
    if decoder_grads:
        # `dec_out`: four tensors the decoder gradients are ACCUMULATED into (the parameters' .grad slices) instead of new ones
        dec = tuple(dec_out) if dec_out is not None else tuple(torch.zeros_like(t) for t in (dec_w0, dec_b0, dec_w1, dec_b1))
        for t, like in zip(dec, (dec_w0, dec_b0, dec_w1, dec_b1)):
            if t.shape != like.shape or t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
                raise RuntimeError("raymarch_bwd: dec_out must be four contiguous fp32 device tensors shaped like the decoder parameters")
        a.d_dec_w0, a.d_dec_b0, a.d_dec_w1, a.d_dec_b1 = (_ptr(t) for t in dec)
    L.check(L.lib().hfagp_raymarch_bwd(C.byref(a), _stream()), "raymarch_bwd")
    if decoder_grads:
        return d_planes, dec
    return (d_planes, rec) if return_rec else d_planes


# ----------------------------------------------------------------------------- loss side of the fitting step
class PoolMSE(torch.autograd.Function):
    """(mean((real - AdaptiveAvgPool2d(size)(img))^2), pooled image) in one pass over `img` [B,C,H,W] (H, W integer
    multiples of real's h, w); backward = one pass writing d img.  The pooled image is returned for the caller's
    book-keeping (the reference returns it from gen_update) and carries no gradient."""

    @staticmethod
    def forward(ctx, img: torch.Tensor, real: torch.Tensor):
        _chk(img, "img"), _chk(real, "real")
        b, c, hh, ww = img.shape
        h, w = real.shape[-2:]
        f = hh // h
        if real.shape[:2] != (b, c) or hh != f * h or ww != f * w or f < 1:
            raise RuntimeError(f"pool_mse: image {tuple(img.shape)} is not an integer multiple of real {tuple(real.shape)}")
        pooled = torch.empty_like(real)
        loss = torch.empty((), device=img.device, dtype=torch.float32)
        ws = torch.empty(L.lib().hfagp_pool_mse_workspace_bytes() // 4, device=img.device, dtype=torch.float32)
        L.check(L.lib().hfagp_pool_mse_fwd(_ptr(img), _ptr(real), _ptr(pooled), _ptr(loss), _ptr(ws), b * c, h, w, f,
                                           _stream()), "pool_mse_fwd")
        ctx.save_for_backward(pooled, real)
        ctx.shape = (b, c, hh, ww, f)
        ctx.mark_non_differentiable(pooled)
        return loss, pooled

    @staticmethod
    def backward(ctx, g_loss, _g_pooled):
        pooled, real = ctx.saved_tensors
        b, c, hh, ww, f = ctx.shape
        d_img = torch.empty(b, c, hh, ww, device=pooled.device, dtype=torch.float32)
        L.check(L.lib().hfagp_pool_mse_bwd(_ptr(pooled), _ptr(real), _ptr(g_loss.contiguous().float()), _ptr(d_img), b * c,
                                           real.shape[-2], real.shape[-1], f, _stream()), "pool_mse_bwd")
        return d_img, None


def pool_mse(img: torch.Tensor, real: torch.Tensor):
    """→ (l2 loss scalar, pooled image [B,C,h,w])."""
    return PoolMSE.apply(img, real)


# ----------------------------------------------------------------------------- optimiser of the fitting step
class AdamTables:
    """Device tables of hfagp_adam_step for one set of (param, grad, exp_avg, exp_avg_sq, step) tensors; built once per set."""

    def __init__(self, params, grads, exp_avgs, exp_avg_sqs, steps):
        chunk = L.lib().hfagp_adam_chunk()
        rows, chunks = [], []
        for k, (p, g, m, v, st) in enumerate(zip(params, grads, exp_avgs, exp_avg_sqs, steps)):
            for t in (p, g, m, v):
                if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
                    raise RuntimeError("adam_step: fp32 contiguous CUDA tensors only")
            n = p.numel()
            rows.append([p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), st.data_ptr(), n])
            chunks.extend([k, e] for e in range(0, n, chunk))
        dev = params[0].device
        self.tensors = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.chunks = torch.tensor(chunks, dtype=torch.int32).to(dev)
        self.n, self.nchunks = len(rows), len(chunks)
        self.keep = (list(params), list(grads), list(exp_avgs), list(exp_avg_sqs), list(steps))     # (the pointers stay valid)


def adam_step(tables: AdamTables, lr: float, beta1: float, beta2: float, eps: float) -> None:
    L.check(L.lib().hfagp_adam_step(_ptr(tables.tensors), _ptr(tables.chunks), tables.n, tables.nchunks, lr, beta1, beta2, eps,
                                    _stream()), "adam_step")

"""`EncoderApp` (the RGB driver's convolutional trunk, /root/reference/code/networks/encoder3d.py:226-260) on the HIP
conv kernels of the generator — forward, data gradients and weight gradients — instead of MIOpen.

The trunk is a StyleGAN2-discriminator-shaped ResNet: per resolution a 3x3 conv, a blurred stride-2 3x3 conv and a
blurred stride-2 1x1 skip.  Every GEMM-shaped piece already exists in `libhfagp_hip.so` for the generator, and the
down-sampling layer is EXACTLY the adjoint of the generator's up-sampling layer:

  Blur(pad 2) + conv3x3(stride 2)   =  hfagp_upfir_bwd (FIR adjoint, gain 4, parity split) + hfagp_modconv_fwd HFAGP_CONVS2_BWD, / 4
  its data gradient                 =  hfagp_modconv_fwd HFAGP_CONVT3X3_UP2 + hfagp_upfir_epilogue_fwd (FIR pad 1, gain 4), / 4
  its weight gradient               =  hfagp_conv_wgrad HFAGP_CONVT3X3_UP2 with the roles of x and g exchanged, / 4
  conv3x3(stride 1) + bias + lrelu  =  hfagp_modconv_fwd HFAGP_CONV3X3 (fused epilogue), HFAGP_CONV3X3_BWD, hfagp_conv_wgrad

(no flips; pinned numerically against `F.conv2d` in tests/test_gpu_round2.py).  Activations are channels-last fp32.
Arithmetic: split-bf16 operands with fp32 accumulation — three parts / six MFMAs per product (2^-23, fp32-class, full fp32
exponent range: the encoder has no clamp and no range tracking) for the forward and data-gradient GEMMs, two parts for the
weight gradients, exact fp32 MFMA for the 64-channel layers the 16-bit kernels do not tile.  The 3-channel input layer, the
1x1 skips, the final 4x4 "conv" (a 8192 -> 512 linear map) and the residual adds are plain torch ops (rocBLAS / elementwise).
Module and parameter names are untouched: this file only provides `EncoderApp.forward` for CUDA tensors.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import ops

SQRT2 = math.sqrt(2.0)
FWD_PREC = "bf16x6"        # forward and data-gradient GEMMs (see the module docstring)


# B-operand images of the trunk's (scaled) weights of the CURRENT step: {(data_ptr, transposed): image}, filled by ONE
# hfagp_weight_prep_batch launch at the start of forward_app — the forward image and the image of the Cin/Cout transpose (the
# data-gradient GEMMs) of every weight from a single read of it.  Per use the step ran 21 split launches + 12 transposing copies.
# The key is the weight's ADDRESS (autograd hands the backward pass a different tensor object over the same storage), so every entry
# holds a reference to its weight: while the entry lives no other tensor can be allocated at that address.
_IMAGES: dict = {}


def _prepare_images(pairs) -> None:
    """pairs: [(scaled weight [Cout,Cin,3,3], its transpose runs an up-sampling kernel?)]"""
    _IMAGES.clear()
    items, keys = [], []
    for w, t_up in pairs:
        cout, cin = w.shape[:2]
        if not (w.is_contiguous() and ops.weight_prep_batch_supported(w)):
            continue
        fwd = FWD_PREC if ops.split_supported(cin, cout, up=False) else None
        bwd = FWD_PREC if ops.split_supported(cout, cin, up=t_up) else None
        if fwd is None and bwd is None:
            continue
        items.append((w, fwd, bwd, False))
        keys.append(w.data_ptr())
    if items:
        for (w, _, _, _), key, (img, img_t, _) in zip(items, keys, ops.weight_prep_batch(items)):
            if img is not None:
                _IMAGES[(key, False)] = (img, w)
            if img_t is not None:
                _IMAGES[(key, True)] = (img_t, w)


def _image(w: torch.Tensor, up: bool = False, transposed: bool = False) -> torch.Tensor:
    """B-operand image of a conv weight [Cout, Cin, k, k] (`transposed`: of its Cin/Cout transpose) for the kernel that will run it."""
    hit = _IMAGES.get((w.data_ptr(), transposed))
    if hit is not None and hit[1].shape == w.shape and hit[1]._version == w._version:
        return hit[0]
    if transposed:
        w = w.transpose(0, 1)
    cout, cin = w.shape[:2]
    w = w.contiguous()
    if ops.split_supported(cin, cout, up=up):
        return ops.weight_prep_prec(w, FWD_PREC)
    return ops.weight_prep(w)[0]


def _wprec(cin: int, cout: int) -> str:
    return "bf16x3" if cin % 64 == 0 and cout % 64 == 0 else "fp32"


class _Conv3x3Act(torch.autograd.Function):
    """y = lrelu(conv3x3(x, w) + b) * sqrt(2), channels-last; the epilogue is fused into the conv kernel."""

    @staticmethod
    def forward(ctx, x, w, bias):
        cout = w.shape[0]
        y = ops.modconv(x, _image(w), cout, ops.CONV3X3, bias=bias.contiguous(), act="lrelu", alpha=0.2, gain=SQRT2)
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        cout, cin = w.shape[:2]
        g = ops.bias_act_bwd(dy.contiguous(), y, "lrelu", 0.2, SQRT2, None)
        db = g.sum((0, 1, 2))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.modconv(g, _image(w, transposed=True), cin, ops.CONV3X3_BWD)
        dw = ops.conv_wgrad(x, None, g, w, ops.CONV3X3, precision=_wprec(cin, cout))
        return dx, dw, db


class _InputConvAct(torch.autograd.Function):
    """First layer, ConvLayer(3, C, 1): y = lrelu(x w^T + b) * sqrt(2) per pixel.  x is the image as channels-last with its 3
    channels padded to 8 (the exact-fp32 conv kernel's K granule): one launch with the fused epilogue instead of a K = 3
    rocBLAS GEMM (0.4 ms) and a separate activation pass; the image needs no gradient."""

    @staticmethod
    def forward(ctx, x4, w, bias):
        cout = w.shape[0]
        w4 = F.pad(w, (0, 0, 0, 0, 0, 5)).contiguous()                      # [C, 3, 1, 1] -> [C, 8, 1, 1]
        y = ops.modconv(x4, ops.weight_prep(w4)[0], cout, ops.CONV1X1, bias=bias.contiguous(), act="lrelu", alpha=0.2,
                        gain=SQRT2)
        ctx.save_for_backward(x4, w4, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x4, w4, y = ctx.saved_tensors
        g = ops.bias_act_bwd(dy.contiguous(), y, "lrelu", 0.2, SQRT2, None)
        dw = ops.conv_wgrad(x4, None, g, w4, ops.CONV1X1)[:, :3]
        return None, dw, g.sum((0, 1, 2))


class _BlurConvDown(torch.autograd.Function):
    """y = conv3x3(4 Blur_pad2(x), w, stride 2), channels-last, no bias / activation (see the module docstring; the caller
    folds the 1/4 into the equalised-lr scale of w)."""

    @staticmethod
    def forward(ctx, x, w):
        cout = w.shape[0]
        gph = ops.upfir_bwd(x)                                   # 4 x blur, as the four parity images (w carries the 1/4)
        y = ops.modconv(gph, _image(w), cout, ops.CONVS2_BWD)
        ctx.save_for_backward(gph, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        gph, w = ctx.saved_tensors
        cout, cin = w.shape[:2]
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            yt = ops.modconv(dy, _image(w, up=True, transposed=True), cin, ops.CONVT3X3_UP2)
            dx = ops.upfir_epilogue(yt, None, None, 0.0, None, "linear", 0.2, 1.0, None)
        like = torch.empty(cin, cout, 3, 3, device=w.device, dtype=torch.float32)
        dw = ops.conv_wgrad(dy, None, gph, like, ops.CONVT3X3_UP2, precision=_wprec(cout, cin)).transpose(0, 1)
        return dx, dw


class _ScaleAll(torch.autograd.Function):
    """(w_1 s_1, ..., w_n s_n) for the equalised-lr scales of every conv of the trunk in ONE multi-tensor launch each way
    (`torch._foreach_mul`) instead of one elementwise kernel per weight forward and another backward (2 x 14 launches of ~5 us at
    size 256: the step is launch-bound there)."""

    @staticmethod
    def forward(ctx, scales, *ws):
        ctx.scales = scales
        return tuple(torch._foreach_mul([w.detach() for w in ws], scales))

    @staticmethod
    def backward(ctx, *gs):
        live = [(i, g) for i, g in enumerate(gs) if g is not None]
        out = [None] * len(gs)
        if live:
            res = torch._foreach_mul([g.contiguous() for _, g in live], [ctx.scales[i] for i, _ in live])
            for (i, _), r in zip(live, res):
                out[i] = r
        return (None, *out)


def supported(net_app, x: torch.Tensor) -> bool:
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[2] == x.shape[3]):
        return False
    if x.requires_grad and torch.is_grad_enabled():
        return False              # a gradient w.r.t. the IMAGE (nobody in HFA-GP asks for one): the plain torch path has it
    size = x.shape[2]
    if size < 16 or size & (size - 1):
        return False
    from .encoder3d import ResBlock
    return all(m.conv1[0].weight.shape[0] % 4 == 0 and m.conv2[1].weight.shape[0] % 4 == 0
               for m in net_app.convs if isinstance(m, ResBlock))


def forward_app(net_app, x: torch.Tensor) -> torch.Tensor:
    """EncoderApp.forward for a CUDA image batch [B,3,size,size] -> [B, w_dim]."""
    from .encoder3d import ResBlock
    convs = list(net_app.convs)
    first, last = convs[0], convs[-1]
    # ---- ConvLayer(3, C, 1): a per-pixel 3 -> C linear map + fused leaky ReLU
    conv, act = first[0], first[1]
    x4 = F.pad(x.detach().permute(0, 2, 3, 1), (0, 5)).contiguous()
    # every equalised-lr weight scale of the trunk in one launch (the down-sampling conv carries the 1/4 of the FIR gain, the
    # skip the 1/sqrt(2) of the residual average)
    blocks = convs[1:-1]
    ws, scales = [conv.weight], [float(conv.scale)]
    for blk in blocks:
        assert isinstance(blk, ResBlock)
        ws += [blk.conv1[0].weight, blk.conv2[1].weight, blk.skip[1].weight]
        scales += [float(blk.conv1[0].scale), float(blk.conv2[1].scale) * 0.25, float(blk.skip[1].scale) / SQRT2]
    ws.append(last.weight)
    scales.append(float(last.scale))
    scaled = _ScaleAll.apply(scales, *ws)
    _prepare_images([(scaled[1 + 3 * k + j], j == 1) for k in range(len(blocks)) for j in (0, 1)])
    h = _InputConvAct.apply(x4, scaled[0], act.bias.reshape(-1))
    for k, blk in enumerate(blocks):
        w1, w2, wsk = scaled[1 + 3 * k: 4 + 3 * k]
        a1 = blk.conv1[1]
        y = _Conv3x3Act.apply(h, w1, a1.bias.reshape(-1))
        a2 = blk.conv2[2]
        y = _BlurConvDown.apply(y, w2)
        # (conv2 + skip) / sqrt(2): the 1/sqrt(2) goes into the activation gain and into the skip's weight scale
        y = ops.bias_act(y, a2.bias.reshape(-1), dim=3, act="lrelu", alpha=a2.negative_slope, gain=a2.scale / SQRT2)
        s = torch.matmul(ops.blur_down(h), wsk[:, :, 0, 0].t())
        h = y + s
    # ---- EqualConv2d(C, w_dim, 4, padding=0) on the 4 x 4 map: one linear map
    out = torch.einsum("byxc,ocyx->bo", h, scaled[-1])
    return out

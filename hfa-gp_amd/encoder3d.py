"""RGB driver network (image → 50 basis coordinates) — counterpart of
/root/reference/code/networks/encoder3d.py:86-298 (`Encoder`, `EncoderApp`, `ResBlock`, `ConvLayer`,
`EqualConv2d`, `EqualLinear`, `Blur`, `FusedLeakyReLU`).

SURVEY.md §8a row D1 leaves this net on PyTorch-ROCm; the module definitions below are that (and the CPU path the golden
vectors pin).  On CUDA/ROCm tensors `EncoderApp.forward` hands the conv trunk to `encoder_hip.py`, which runs it on the
generator's HIP conv kernels (the down-sampling layer is the adjoint of the generator's up-sampling layer).  What matters for drop-in use is
that module/parameter NAMES match the reference so that HFA-GP checkpoints load
(`encoder.net_app.convs.N...`, `encoder.fc.N.{weight,bias}`) and that outputs match the golden
vectors captured from the reference (tests/test_host_golden.py).
"""
from __future__ import annotations

import math
from typing import Sequence

import torch
import torch.nn.functional as F
from torch import nn

SQRT2 = math.sqrt(2.0)


def fused_leaky_relu(x: torch.Tensor, bias: torch.Tensor, negative_slope: float = 0.2, scale: float = SQRT2):
    if x.is_cuda and x.dtype == torch.float32 and x.dim() >= 2 and bias.numel() == x.shape[1]:
        # one HIP pass forward (hfagp_bias_act_fwd) and one backward (hfagp_bias_act_bwd) instead of add / leaky_relu / mul
        from . import ops
        return ops.bias_act(x.contiguous(), bias.reshape(-1), dim=1, act="lrelu", alpha=negative_slope, gain=scale)
    return F.leaky_relu(x + bias, negative_slope) * scale


def make_kernel(taps: Sequence[float]) -> torch.Tensor:
    k = torch.as_tensor(taps, dtype=torch.float32)
    if k.ndim == 1:
        k = torch.outer(k, k)
    return k / k.sum()


def upfirdn2d(x: torch.Tensor, kernel: torch.Tensor, up: int = 1, down: int = 1, pad=(0, 0)) -> torch.Tensor:
    """Zero-insert by `up`, pad (negative = crop) by pad=(before, after) on both axes, correlate with the
    flipped kernel, keep every `down`-th sample.  NCHW, pure PyTorch (CPU or ROCm)."""
    n, c, h, w = x.shape
    p0, p1 = pad
    if x.is_cuda and x.dtype == torch.float32:
        # the HIP FIR (hfagp_upfirdn2d_fwd / _bwd; pinned to the reference's upfirdn2d_native by tests/golden) instead of
        # zero-insert + pad + crop + a one-channel F.conv2d over n*c images (which MIOpen serves with its naive kernel)
        from . import ops
        return ops.upfirdn2d(x.contiguous(), kernel.contiguous(), up=up, down=down, padding=(p0, p1, p0, p1), gain=1.0)
    if up > 1:
        z = x.new_zeros(n, c, h, up, w, up)
        z[:, :, :, 0, :, 0] = x
        x = z.view(n, c, h * up, w * up)
    x = F.pad(x, [max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)])
    x = x[:, :, max(-p0, 0): x.shape[2] - max(-p1, 0), max(-p0, 0): x.shape[3] - max(-p1, 0)]
    kh, kw = kernel.shape
    wgt = torch.flip(kernel, [0, 1]).view(1, 1, kh, kw).to(x.dtype)
    y = F.conv2d(x.reshape(n * c, 1, x.shape[2], x.shape[3]), wgt)
    y = y.view(n, c, y.shape[2], y.shape[3])
    return y[:, :, ::down, ::down]


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel: int, negative_slope: float = 0.2, scale: float = SQRT2):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(1, channel, 1, 1))
        self.negative_slope, self.scale = negative_slope, scale

    def forward(self, x):
        return fused_leaky_relu(x, self.bias, self.negative_slope, self.scale)


class ScaledLeakyReLU(nn.Module):
    def __init__(self, negative_slope: float = 0.2):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, x):
        return F.leaky_relu(x, negative_slope=self.negative_slope)


class Blur(nn.Module):
    def __init__(self, kernel, pad, upsample_factor: int = 1):
        super().__init__()
        k = make_kernel(kernel)
        if upsample_factor > 1:
            k = k * (upsample_factor ** 2)
        self.register_buffer("kernel", k)
        self.pad = pad

    def forward(self, x):
        return upfirdn2d(x, self.kernel, pad=self.pad)


class EqualConv2d(nn.Module):
    """Conv with N(0,1) weights and the 1/sqrt(fan_in) equalised-lr scale applied at run time."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride, self.padding = stride, padding
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None

    def forward(self, x):
        return F.conv2d(x, self.weight * self.scale, bias=self.bias, stride=self.stride, padding=self.padding)


class _EqualLinearFn(torch.autograd.Function):
    """y = x @ (W * scale)^T + b * lr_mul with the scale folded into the GEMM calls (`alpha`): the plain expression costs two
    scalar-multiply kernels per layer in the forward pass and two more in autograd's backward — 28 launches of ~3 us per step
    for the 3DMM driver's seven-layer chain, a third of its kernels."""

    @staticmethod
    def forward(ctx, x, weight, bias, scale, lr_mul):
        ctx.save_for_backward(x, weight)
        ctx.scale, ctx.lr_mul = scale, lr_mul
        return torch.addmm(bias if lr_mul == 1 else bias * lr_mul, x, weight.t(), alpha=scale)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.addmm(g.new_empty(x.shape), g, weight, beta=0, alpha=ctx.scale)
        if ctx.needs_input_grad[1]:
            gw = torch.addmm(g.new_empty(weight.shape), g.t(), x, beta=0, alpha=ctx.scale)
        if ctx.needs_input_grad[2]:
            gb = g.sum(0) if ctx.lr_mul == 1 else g.sum(0) * ctx.lr_mul
        return gx, gw, gb, None, None


class EqualLinear(nn.Module):
    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, x):
        if self.activation:
            return fused_leaky_relu(F.linear(x, self.weight * self.scale), self.bias * self.lr_mul)
        if x.dim() == 2 and self.bias is not None and x.dtype == self.weight.dtype:
            return _EqualLinearFn.apply(x, self.weight, self.bias, self.scale, self.lr_mul)
        return F.linear(x, self.weight * self.scale, bias=self.bias * self.lr_mul)


class ConvLayer(nn.Sequential):
    """[Blur] → EqualConv2d → [FusedLeakyReLU | ScaledLeakyReLU]; stride-2 variant blurs first."""

    def __init__(self, in_channel, out_channel, kernel_size, downsample=False, blur_kernel=(1, 3, 3, 1),
                 bias=True, activate=True):
        mods = []
        if downsample:
            p = (len(blur_kernel) - 2) + (kernel_size - 1)
            mods.append(Blur(blur_kernel, pad=((p + 1) // 2, p // 2)))
            stride, self.padding = 2, 0
        else:
            stride, self.padding = 1, kernel_size // 2
        mods.append(EqualConv2d(in_channel, out_channel, kernel_size, padding=self.padding, stride=stride,
                                bias=bias and not activate))
        if activate:
            mods.append(FusedLeakyReLU(out_channel) if bias else ScaledLeakyReLU(0.2))
        super().__init__(*mods)


class ResBlock(nn.Module):
    def __init__(self, in_channel, out_channel, blur_kernel=(1, 3, 3, 1)):
        super().__init__()
        self.conv1 = ConvLayer(in_channel, in_channel, 3)
        self.conv2 = ConvLayer(in_channel, out_channel, 3, downsample=True)
        self.skip = ConvLayer(in_channel, out_channel, 1, downsample=True, activate=False, bias=False)

    def forward(self, x):
        return (self.conv2(self.conv1(x)) + self.skip(x)) / SQRT2


_CHANNELS = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256, 128: 128, 256: 64, 512: 32, 1024: 16}


class EncoderApp(nn.Module):
    def __init__(self, size: int, w_dim: int = 512):
        super().__init__()
        self.w_dim = w_dim
        log_size = int(math.log(size, 2))
        convs = [ConvLayer(3, _CHANNELS[size], 1)]
        cin = _CHANNELS[size]
        for i in range(log_size, 2, -1):
            cout = _CHANNELS[2 ** (i - 1)]
            convs.append(ResBlock(cin, cout))
            cin = cout
        convs.append(EqualConv2d(cin, w_dim, 4, padding=0, bias=False))
        self.convs = nn.ModuleList(convs)

    def forward(self, x):
        if x.is_cuda:
            # CUDA/ROCm tensors: the trunk runs on the HIP conv kernels (encoder_hip.py) where its shapes allow
            from . import encoder_hip
            if encoder_hip.supported(self, x):
                return encoder_hip.forward_app(self, x)
        for m in self.convs:
            x = m(x)
        return x.squeeze(-1).squeeze(-1)


def _mlp(dims) -> nn.Sequential:
    return nn.Sequential(*[EqualLinear(a, b) for a, b in zip(dims[:-1], dims[1:])])


class Encoder(nn.Module):
    """image [B,3,size,size] → basis coordinates [B, dim_motion] (five activation-less EqualLinear),
    optionally a 25-float pose head."""

    def __init__(self, size, dim=512, dim_motion=20, use_softmax=False, out_pose=False):
        super().__init__()
        self.net_app = EncoderApp(size, dim)
        self.fc = _mlp([dim] * 5 + [dim_motion])
        self.out_pose = out_pose
        if out_pose:
            self.pose = _mlp([dim] * 5 + [25])
        self.use_softmax = use_softmax
        self.softmax = nn.Softmax(dim=1)

    def enc_app(self, x):
        return self.net_app(x)

    def get_weights(self, x):
        h = self.net_app(x)
        w = self.fc(h)
        if self.use_softmax:
            w = self.softmax(w)
        if self.out_pose:
            return w, self.pose(h)
        return w

    def forward(self, input_source, h_start=None):
        return self.get_weights(input_source)

"""CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE) for the EG3D tri-plane generator.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path (``hfa_gp_amd``) never does and
fails loudly when its HIP library is missing.

PARITY UNPINNED for everything in this file: the algorithm lives in a
third-party dependency, NVlabs/eg3d (modules ``training.triplane``,
``training.networks_stylegan2``, ``training.superresolution``,
``training.volumetric_rendering.{renderer,ray_sampler,ray_marcher}``,
``torch_utils.ops.{bias_act,upfirdn2d,conv2d_resample}``), which is absent from
``/root/reference`` and has no pinned version there (no requirements file; the
implicit pin is the code pickled in ``ffhqrebalanced512-128.pkl``, which is not
shipped either).  The reference reaches it only through
``generator.synthesis(ws, c=label, noise_mode='const')['image']``
(/root/reference/code/networks/headnerf.py:112,118,133,207,218,267,277) and
holds no test or golden vector at that boundary.  What follows restates the
published EG3D algorithm in plain fp32
PyTorch with the same ATen ops its CPU/`ref` path uses; the pieces that CAN be
pinned are pinned in tests/: FIR resampling against the reference's own
``upfirdn2d_native`` (code/networks/encoder3d.py:23-45, via golden vectors),
bias+lrelu against ``fused_leaky_relu`` (encoder3d.py:7-8), bilinear sampling
against ``F.grid_sample``, and closed-form compositing cases.

All functions are functional: parameters come in a flat ``dict`` keyed with
EG3D ``state_dict`` names (``backbone.synthesis.b8.conv0.weight`` ...).
The renderer's random draws (EG3D: ``torch.rand_like`` for the stratified
jitter, ``torch.rand`` in ``sample_pdf``) are explicit inputs ``u_strat`` /
``u_imp`` so that GPU-vs-oracle parity is defined.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# §11.1  FIR resampling (EG3D torch_utils/ops/upfirdn2d.py `_upfirdn2d_ref`;
# same convention as the reference's upfirdn2d_native, encoder3d.py:23-45)
# --------------------------------------------------------------------------
def fir_kernel(taps=(1, 3, 3, 1)) -> Tensor:
    k = torch.tensor(taps, dtype=torch.float32)
    k = torch.outer(k, k)
    return k / k.sum()


def upfirdn2d(x: Tensor, f: Tensor, up: int = 1, down: int = 1,
              padding=(0, 0, 0, 0), gain: float = 1.0) -> Tensor:
    """x [N,C,H,W]; padding = (px0, px1, py0, py1); zero-insert, pad/crop,
    correlate with flipped f * gain, decimate."""
    n, c, h, w = x.shape
    px0, px1, py0, py1 = padding
    if up > 1:
        z = x.new_zeros(n, c, h, up, w, up)
        z[:, :, :, 0, :, 0] = x
        x = z.reshape(n, c, h * up, w * up)
    x = F.pad(x, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    x = x[:, :, max(-py0, 0): x.shape[2] - max(-py1, 0), max(-px0, 0): x.shape[3] - max(-px1, 0)]
    k = (f * gain).to(x.dtype).flip([0, 1])[None, None].repeat(c, 1, 1, 1)
    x = F.conv2d(x, k, groups=c)
    return x[:, :, ::down, ::down]


def upsample2d(x: Tensor, f: Tensor) -> Tensor:
    """Skip-image path: up=2, padding [2,1,2,1], gain 4 (§11.1)."""
    return upfirdn2d(x, f, up=2, padding=(2, 1, 2, 1), gain=4.0)


# --------------------------------------------------------------------------
# bias_act (EG3D torch_utils/ops/bias_act.py `_bias_act_ref`)
# --------------------------------------------------------------------------
def bias_act(x: Tensor, b: Optional[Tensor], act: str = "linear", alpha: float = 0.2,
             gain: Optional[float] = None, clamp: Optional[float] = None, dim: int = 1) -> Tensor:
    def_gain = math.sqrt(2.0) if act == "lrelu" else 1.0
    gain = def_gain if gain is None else gain
    if b is not None:
        shape = [1] * x.ndim
        shape[dim] = -1
        x = x + b.reshape(shape)
    if act == "lrelu":
        x = F.leaky_relu(x, alpha)
    elif act != "linear":
        raise ValueError(act)
    if gain != 1.0:
        x = x * gain
    if clamp is not None and clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


# --------------------------------------------------------------------------
# §11.5  FullyConnectedLayer
# --------------------------------------------------------------------------
def fully_connected(x: Tensor, weight: Tensor, bias: Optional[Tensor], lr_mul: float = 1.0) -> Tensor:
    w = weight * (lr_mul / math.sqrt(weight.shape[1]))
    y = x @ w.t()
    if bias is not None:
        y = y + bias * lr_mul
    return y


# --------------------------------------------------------------------------
# §11.2 / §11.3  modulated convolution (both algebraic forms)
# --------------------------------------------------------------------------
def _conv_up2(x: Tensor, w: Tensor, f: Tensor, groups: int = 1) -> Tensor:
    """conv2d_resample fast path for up=2, 3x3, layer padding 1:
    conv_transpose2d(stride 2, padding 0, un-flipped weight) then
    upfirdn2d(f, padding [1,1,1,1], gain 4)."""
    o, i, kh, kw = w.shape
    if groups == 1:
        wt = w.transpose(0, 1)
    else:
        wt = w.reshape(groups, o // groups, i, kh, kw).transpose(1, 2).reshape(groups * i, o // groups, kh, kw)
    y = F.conv_transpose2d(x, wt, stride=2, padding=0, groups=groups)
    return upfirdn2d(y, f, padding=(1, 1, 1, 1), gain=4.0)


def modulated_conv2d(x: Tensor, weight: Tensor, styles: Tensor, noise: Optional[Tensor] = None,
                     up: int = 1, f: Optional[Tensor] = None, demodulate: bool = True,
                     fused: bool = True, eps: float = 1e-8) -> Tensor:
    b, cin = styles.shape
    cout, _, kh, kw = weight.shape
    pad = kh // 2
    wmod = weight[None] * styles[:, None, :, None, None]            # [B,O,I,k,k]
    dcoef = (wmod.square().sum(dim=[2, 3, 4]) + eps).rsqrt() if demodulate else None
    if not fused:
        x = x * styles[:, :, None, None]
        y = _conv_up2(x, weight, f) if up == 2 else F.conv2d(x, weight, padding=pad)
        if demodulate:
            y = y * dcoef[:, :, None, None]
        if noise is not None:
            y = y + noise
        return y
    if demodulate:
        wmod = wmod * dcoef[:, :, None, None, None]
    xg = x.reshape(1, b * cin, *x.shape[2:])
    wg = wmod.reshape(b * cout, cin, kh, kw)
    y = _conv_up2(xg, wg, f, groups=b) if up == 2 else F.conv2d(xg, wg, padding=pad, groups=b)
    y = y.reshape(b, cout, *y.shape[2:])
    if noise is not None:
        y = y + noise
    return y


def synthesis_layer(x: Tensor, w: Tensor, P: Dict[str, Tensor], pre: str, up: int, f: Tensor,
                    noise_mode: str, conv_clamp: Optional[float], alpha: float, fused: bool,
                    eps: float) -> Tensor:
    styles = fully_connected(w, P[pre + ".affine.weight"], P[pre + ".affine.bias"])
    noise = None
    if noise_mode == "const":
        noise = P[pre + ".noise_const"] * P[pre + ".noise_strength"]
    y = modulated_conv2d(x, P[pre + ".weight"], styles, noise=noise, up=up, f=f, fused=fused, eps=eps)
    return bias_act(y, P[pre + ".bias"], act="lrelu", alpha=alpha, clamp=conv_clamp)


def torgb_layer(x: Tensor, w: Tensor, P: Dict[str, Tensor], pre: str,
                conv_clamp: Optional[float], fused: bool) -> Tensor:
    cin = P[pre + ".weight"].shape[1]
    styles = fully_connected(w, P[pre + ".affine.weight"], P[pre + ".affine.bias"]) * (1.0 / math.sqrt(cin))
    y = modulated_conv2d(x, P[pre + ".weight"], styles, demodulate=False, fused=fused)
    return bias_act(y, P[pre + ".bias"], clamp=conv_clamp)


def synthesis_block(x: Optional[Tensor], img: Optional[Tensor], ws: Tensor, P: Dict[str, Tensor],
                    pre: str, first: bool, f: Tensor, noise_mode: str, conv_clamp: Optional[float],
                    alpha: float, fused: bool, eps: float) -> Tuple[Tensor, Tensor]:
    """EG3D SynthesisBlock, architecture 'skip' (§3.4 step 3 / step 6)."""
    it = iter(ws.unbind(1))
    if first:
        x = P[pre + ".const"][None].repeat(ws.shape[0], 1, 1, 1)
        x = synthesis_layer(x, next(it), P, pre + ".conv1", 1, f, noise_mode, conv_clamp, alpha, fused, eps)
    else:
        x = synthesis_layer(x, next(it), P, pre + ".conv0", 2, f, noise_mode, conv_clamp, alpha, fused, eps)
        x = synthesis_layer(x, next(it), P, pre + ".conv1", 1, f, noise_mode, conv_clamp, alpha, fused, eps)
    if img is not None:
        img = upsample2d(img, f)
    y = torgb_layer(x, next(it), P, pre + ".torgb", conv_clamp, fused)
    img = y if img is None else img + y
    return x, img


def backbone_synthesis(P: Dict[str, Tensor], cfg, ws: Tensor, fused: bool = True) -> Tensor:
    """StyleGAN2 SynthesisNetwork.forward → planes [B, 96, R, R] (§8a A6, §10 U9)."""
    f = fir_kernel(cfg.resample_filter)
    x = img = None
    idx = 0
    for res in cfg.block_resolutions:
        n_conv = 1 if res == 4 else 2
        cur = ws[:, idx: idx + n_conv + 1]
        idx += n_conv
        x, img = synthesis_block(x, img, cur, P, f"backbone.synthesis.b{res}", res == 4, f,
                                 cfg.backbone_noise_mode, cfg.backbone_conv_clamp, cfg.lrelu_alpha,
                                 fused, cfg.demod_eps)
    return img


def superresolution(P: Dict[str, Tensor], cfg, rgb: Tensor, x: Tensor, ws: Tensor, fused: bool = True) -> Tensor:
    """SuperresolutionHybrid8XDC.forward (§8a A11, §10 U10)."""
    f = fir_kernel(cfg.resample_filter)
    w3 = ws[:, -1:, :].repeat(1, 3, 1)
    x, rgb = synthesis_block(x, rgb, w3, P, "superresolution.block0", False, f, cfg.sr_noise_mode,
                             cfg.sr_conv_clamp, cfg.lrelu_alpha, fused, cfg.demod_eps)
    x, rgb = synthesis_block(x, rgb, w3, P, "superresolution.block1", False, f, cfg.sr_noise_mode,
                             cfg.sr_conv_clamp, cfg.lrelu_alpha, fused, cfg.demod_eps)
    return rgb


# --------------------------------------------------------------------------
# §11.9  ray generation (EG3D RaySampler.forward)
# --------------------------------------------------------------------------
def ray_sampler(c2w: Tensor, intr: Tensor, res: int) -> Tuple[Tensor, Tensor]:
    n = c2w.shape[0]
    cam = c2w[:, :3, 3]
    fx, fy = intr[:, 0, 0], intr[:, 1, 1]
    cx, cy = intr[:, 0, 2], intr[:, 1, 2]
    sk = intr[:, 0, 1]
    ar = torch.arange(res, dtype=c2w.dtype)      # (fp32 in EG3D; the tests also run this oracle in fp64 as ground truth)
    uv = torch.stack(torch.meshgrid(ar, ar, indexing="ij")) * (1.0 / res) + (0.5 / res)
    uv = uv.flip(0).reshape(2, -1).transpose(1, 0)[None].repeat(n, 1, 1)
    xc, yc = uv[:, :, 0], uv[:, :, 1]
    zc = torch.ones_like(xc)
    xl = (xc - cx[:, None] + cy[:, None] * sk[:, None] / fy[:, None] - sk[:, None] * yc / fy[:, None]) / fx[:, None] * zc
    yl = (yc - cy[:, None]) / fy[:, None] * zc
    pts = torch.stack((xl, yl, zc, torch.ones_like(zc)), dim=-1)
    world = torch.bmm(c2w, pts.permute(0, 2, 1)).permute(0, 2, 1)[:, :, :3]
    dirs = F.normalize(world - cam[:, None, :], dim=2)
    origins = cam[:, None, :].repeat(1, dirs.shape[1], 1)
    return origins, dirs


# --------------------------------------------------------------------------
# §11.6  tri-plane sampling
# --------------------------------------------------------------------------
def plane_axes(kind: str) -> Tensor:
    third = [[0, 0, 1], [1, 0, 0], [0, 1, 0]] if kind == "eg3d_original" else [[0, 0, 1], [0, 1, 0], [1, 0, 0]]
    return torch.tensor([[[1, 0, 0], [0, 1, 0], [0, 0, 1]],
                         [[1, 0, 0], [0, 0, 1], [0, 1, 0]],
                         third], dtype=torch.float32)


def sample_from_planes(axes: Tensor, planes: Tensor, coords: Tensor, box_warp: float) -> Tensor:
    """planes [N,3,C,H,W], coords [N,M,3] → [N,3,M,C]."""
    n, npl, c, h, w = planes.shape
    m = coords.shape[1]
    coords = (2.0 / box_warp) * coords
    cexp = coords[:, None].expand(-1, npl, -1, -1).reshape(n * npl, m, 3)
    inv = torch.linalg.inv(axes).to(coords.dtype)[None].expand(n, -1, -1, -1).reshape(n * npl, 3, 3)
    proj = torch.bmm(cexp, inv)[..., :2]
    out = F.grid_sample(planes.reshape(n * npl, c, h, w), proj[:, None].to(planes.dtype), mode="bilinear",
                        padding_mode="zeros", align_corners=False)
    return out.permute(0, 3, 2, 1).reshape(n, npl, m, c)


def osg_decoder(P: Dict[str, Tensor], feats: Tensor, lr_mul: float) -> Tuple[Tensor, Tensor]:
    """feats [N,3,M,32] → rgb [N,M,32], sigma [N,M,1] (§8a A8)."""
    x = feats.mean(1)
    n, m, c = x.shape
    x = x.reshape(n * m, c)
    x = fully_connected(x, P["decoder.net.0.weight"], P["decoder.net.0.bias"], lr_mul)
    x = F.softplus(x)
    x = fully_connected(x, P["decoder.net.2.weight"], P["decoder.net.2.bias"], lr_mul)
    x = x.reshape(n, m, -1)
    rgb = torch.sigmoid(x[..., 1:]) * (1 + 2 * 0.001) - 0.001
    return rgb, x[..., 0:1]


# --------------------------------------------------------------------------
# §11.8  MipRayMarcher2
# --------------------------------------------------------------------------
def ray_march(colors: Tensor, dens: Tensor, depths: Tensor, white_back: bool = False):
    deltas = depths[:, :, 1:] - depths[:, :, :-1]
    c_mid = (colors[:, :, :-1] + colors[:, :, 1:]) / 2
    d_mid = (dens[:, :, :-1] + dens[:, :, 1:]) / 2
    t_mid = (depths[:, :, :-1] + depths[:, :, 1:]) / 2
    d_mid = F.softplus(d_mid - 1)
    alpha = 1 - torch.exp(-(d_mid * deltas))
    shifted = torch.cat([torch.ones_like(alpha[:, :, :1]), 1 - alpha + 1e-10], -2)
    weights = alpha * torch.cumprod(shifted, -2)[:, :, :-1]
    rgb = torch.sum(weights * c_mid, -2)
    wtot = weights.sum(2)
    depth = torch.sum(weights * t_mid, -2) / wtot
    depth = torch.nan_to_num(depth, float("inf"))
    depth = torch.clamp(depth, torch.min(depths), torch.max(depths))
    if white_back:
        rgb = rgb + 1 - wtot
    rgb = rgb * 2 - 1
    return rgb, depth, weights


# --------------------------------------------------------------------------
# §11.7  depths: stratified + importance
# --------------------------------------------------------------------------
def sample_stratified(n: int, m: int, start: float, end: float, s: int, u: Tensor) -> Tensor:
    d = torch.linspace(start, end, s).reshape(1, 1, s, 1).repeat(n, m, 1, 1)
    delta = (end - start) / (s - 1)
    return d + u * delta


def sample_pdf(bins: Tensor, weights: Tensor, u: Tensor, eps: float = 1e-5) -> Tensor:
    nr, ns = weights.shape
    weights = weights + eps
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp_min(inds - 1, 0)
    above = torch.clamp_max(inds, ns)
    ig = torch.stack([below, above], -1).view(nr, -1)
    cdf_g = torch.gather(cdf, 1, ig).view(nr, -1, 2)
    bins_g = torch.gather(bins, 1, ig).view(nr, -1, 2)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom = torch.where(denom < eps, torch.ones_like(denom), denom)
    return bins_g[..., 0] + (u - cdf_g[..., 0]) / denom * (bins_g[..., 1] - bins_g[..., 0])


def sample_importance(z: Tensor, weights: Tensor, u_imp: Tensor) -> Tensor:
    """EG3D runs this under torch.no_grad() and detaches the result: the importance depths carry no
    gradient (neither to the densities nor to the planes)."""
    with torch.no_grad():
        b, r, s, _ = z.shape
        z = z.reshape(b * r, s)
        w = weights.reshape(b * r, -1)
        w = F.max_pool1d(w[:, None].to(torch.promote_types(w.dtype, torch.float32)), 2, 1, padding=1)   # EG3D: .float()
        w = F.avg_pool1d(w, 2, 1).squeeze(1)
        w = w + 0.01
        z_mid = 0.5 * (z[:, :-1] + z[:, 1:])
        out = sample_pdf(z_mid, w[:, 1:-1], u_imp.reshape(b * r, -1))
    return out.detach().reshape(b, r, -1, 1)


def importance_renderer(P: Dict[str, Tensor], cfg, planes: Tensor, origins: Tensor, dirs: Tensor,
                        u_strat: Tensor, u_imp: Tensor, fine_depths: Optional[Tensor] = None):
    """ImportanceRenderer.forward (§3.4 step 4). planes [B,3,32,H,W];
    u_strat [B,R,S,1]; u_imp [B*R,Sf] → feat [B,R,32], depth [B,R,1], wsum [B,R,1].
    ``fine_depths`` [B,R,Sf,1] (tests only): importance depths given from outside instead of drawn from the coarse
    pass — both sides of a comparison then sample the volume at the SAME points, which separates arithmetic error
    from the sensitivity of the inverse-CDF sampling to last-bit differences of the coarse weights."""
    axes = plane_axes(cfg.plane_axes)
    b, r, _ = origins.shape
    s = cfg.depth_resolution

    def run(depths):
        xyz = (origins[:, :, None] + depths * dirs[:, :, None]).reshape(b, -1, 3)
        feats = sample_from_planes(axes, planes, xyz, cfg.box_warp)
        rgb, sigma = osg_decoder(P, feats, cfg.decoder_lr_mul)
        k = depths.shape[2]
        return rgb.reshape(b, r, k, -1), sigma.reshape(b, r, k, 1)

    d_c = sample_stratified(b, r, cfg.ray_start, cfg.ray_end, s, u_strat)
    c_c, s_c = run(d_c)
    if cfg.depth_resolution_importance > 0:
        _, _, w = ray_march(c_c, s_c, d_c, cfg.white_back)
        d_f = sample_importance(d_c, w, u_imp) if fine_depths is None else fine_depths.detach()
        c_f, s_f = run(d_f)
        d_all = torch.cat([d_c, d_f], -2)
        c_all = torch.cat([c_c, c_f], -2)
        s_all = torch.cat([s_c, s_f], -2)
        _, idx = torch.sort(d_all, dim=-2)
        d_all = torch.gather(d_all, -2, idx)
        c_all = torch.gather(c_all, -2, idx.expand(-1, -1, -1, c_all.shape[-1]))
        s_all = torch.gather(s_all, -2, idx.expand(-1, -1, -1, 1))
        rgb, depth, w = ray_march(c_all, s_all, d_all, cfg.white_back)
    else:
        rgb, depth, w = ray_march(c_c, s_c, d_c, cfg.white_back)
    return rgb, depth, w.sum(2)


# --------------------------------------------------------------------------
# §3.4  TriPlaneGenerator.synthesis
# --------------------------------------------------------------------------
def synthesis(P: Dict[str, Tensor], cfg, ws: Tensor, c: Tensor, u_strat: Tensor, u_imp: Tensor,
              fused: bool = True, return_planes: bool = False, fine_depths: Optional[Tensor] = None) -> Dict[str, Tensor]:
    b = ws.shape[0]
    c2w = c[:, :16].reshape(-1, 4, 4)
    intr = c[:, 16:25].reshape(-1, 3, 3)
    res = cfg.neural_rendering_resolution
    origins, dirs = ray_sampler(c2w, intr, res)
    planes = backbone_synthesis(P, cfg, ws, fused)
    planes5 = planes.reshape(b, 3, cfg.plane_channels, planes.shape[-2], planes.shape[-1])
    feat, depth, _ = importance_renderer(P, cfg, planes5, origins, dirs, u_strat, u_imp, fine_depths)
    feat_img = feat.permute(0, 2, 1).reshape(b, feat.shape[-1], res, res).contiguous()
    depth_img = depth.permute(0, 2, 1).reshape(b, 1, res, res)
    rgb_raw = feat_img[:, :3]
    img = superresolution(P, cfg, rgb_raw, feat_img, ws, fused)
    out = {"image": img, "image_raw": rgb_raw, "image_depth": depth_img}
    if return_planes:
        out["planes"] = planes
        out["feature_image"] = feat_img
    return out


def mapping(P: Dict[str, Tensor], cfg, z: Tensor, c: Tensor, truncation_psi: float = 1.0) -> Tensor:
    """MappingNetwork.forward (never called by HFA-GP)."""
    def norm2(x):
        return x * (x.square().mean(1, keepdim=True) + 1e-8).rsqrt()
    x = norm2(z)
    y = norm2(fully_connected(c, P["backbone.mapping.embed.weight"], P["backbone.mapping.embed.bias"]))
    x = torch.cat([x, y], 1)
    for i in range(cfg.mapping_layers):
        x = fully_connected(x, P[f"backbone.mapping.fc{i}.weight"], P[f"backbone.mapping.fc{i}.bias"],
                            cfg.mapping_lr_mul)
        x = F.leaky_relu(x, 0.2) * math.sqrt(2.0)
    ws = x[:, None].repeat(1, cfg.num_ws, 1)
    if truncation_psi != 1.0:
        ws = P["backbone.mapping.w_avg"].lerp(ws, truncation_psi)
    return ws

"""Backward pass of the HIP path (d/d ws with the generator frozen) against autograd through the CPU oracle.
Needs an MI355X:  python -m pytest tests -m gpu"""
import dataclasses
import math

import pytest
import torch
import torch.nn.functional as F

from tests.util import look_at_label, make_inputs, perturb_state, state_cpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("the -m gpu tests need an MI355X")
    from hfa_gp_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def close(a, b, atol, rtol=1e-4):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs()
    assert torch.isfinite(a).all()
    assert bool((err <= atol + rtol * b.abs()).all()), f"max err {err.max().item():.3e} (ref max {b.abs().max().item():.3e})"


def _gemm_image_t(ops, w, prec):
    """B-operand image of the Cin/Cout-transposed weight in the layout of ``prec``."""
    wt = w.transpose(0, 1).contiguous()
    return ops.weight_prep(wt)[0] if prec == "fp32" else ops.weight_prep_split(wt, 2 if prec == "bf16x3" else 3)


BWD_TOL = {"fp32": 1e-6, "bf16x3": 5e-5, "bf16x6": 4e-6}     # relative to max|ref|


@pytest.mark.parametrize("b,h,cin,cout,prec", [(1, 4, 8, 32, "fp32"), (2, 9, 16, 24, "fp32"), (1, 17, 32, 128, "fp32"),
                                               (2, 17, 128, 32, "bf16x3"), (1, 9, 256, 48, "bf16x6")])
def test_conv3x3_bwd_data(dev, b, h, cin, cout, prec):
    """mode CONV3X3_BWD with transposed weights == autograd of F.conv2d w.r.t. its input."""
    from hfa_gp_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(b, cin, h, h, generator=g, requires_grad=True)
    w = torch.randn(cout, cin, 3, 3, generator=g)
    gy = torch.randn(b, cout, h, h, generator=g)
    F.conv2d(x, w, padding=1).backward(gy)
    dx = ops.modconv(ops.nchw_to_nhwc(gy.to(dev)), _gemm_image_t(ops, w.to(dev), prec), cin, ops.CONV3X3_BWD)
    close(ops.nhwc_to_nchw(dx), x.grad, atol=2e-5 + BWD_TOL[prec] * float(x.grad.abs().max()))


@pytest.mark.parametrize("b,h,cin,cout,prec", [(1, 4, 8, 32, "fp32"), (2, 7, 16, 24, "fp32"), (1, 16, 32, 64, "fp32"),
                                               (2, 7, 128, 16, "bf16x3"), (1, 16, 128, 64, "bf16x6")])
def test_upconv_bwd_data(dev, b, h, cin, cout, prec):
    """upfir_bwd + mode CONVS2_BWD == autograd of (conv_transpose2d stride 2 -> FIR pad 1 gain 4) w.r.t. input."""
    from hfa_gp_amd import ops
    from oracle import eg3d_oracle as O
    g = torch.Generator().manual_seed(2)
    x = torch.randn(b, cin, h, h, generator=g, requires_grad=True)
    w = torch.randn(cout, cin, 3, 3, generator=g)
    gy = torch.randn(b, cout, 2 * h, 2 * h, generator=g)
    O._conv_up2(x, w, O.fir_kernel()).backward(gy)
    gph = ops.upfir_bwd(ops.nchw_to_nhwc(gy.to(dev)))
    dx = ops.modconv(gph, _gemm_image_t(ops, w.to(dev), prec), cin, ops.CONVS2_BWD)
    close(ops.nhwc_to_nchw(dx), x.grad, atol=5e-5 + BWD_TOL[prec] * float(x.grad.abs().max()))


def test_upsample2d_bwd(dev):
    from hfa_gp_amd import ops
    from oracle import eg3d_oracle as O
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 5, 7, generator=g, requires_grad=True)
    gy = torch.randn(2, 8, 10, 14, generator=g)
    O.upsample2d(x, O.fir_kernel()).backward(gy)
    close(ops.upsample2d_bwd(gy.to(dev), channels_last=False), x.grad, atol=1e-5)
    gh = ops.nchw_to_nhwc(gy.to(dev))
    close(ops.nhwc_to_nchw(ops.upsample2d_bwd(gh, channels_last=True)), x.grad, atol=1e-5)


@pytest.mark.parametrize("preset,axes,ph,pw,box_warp", [
    ("tiny64", "eg3d_original", 20, 20, None), ("small128", "eg3d_original", 20, 20, None),
    ("ffhq512_128", "eg3d_original", 20, 20, None), ("ffhq512_128", "eg3d_fixed", 20, 20, None),
    # VERDICT r5 #6a: the edge shapes of the sort + gather backward against the ORACLE's autograd, not against the scatter kernels —
    # non-square planes (rows != columns of the 32-texel strips, mirrored plane of a non-square plane), a box so small that most
    # bilinear taps fall outside the planes (zeros padding), both axis conventions
    ("small128", "eg3d_original", 20, 36, None), ("small128", "eg3d_fixed", 24, 24, 0.45),
    ("small128", "eg3d_original", 40, 72, 0.6), ("small128", "eg3d_fixed", 36, 20, 0.45)])
def test_raymarch_bwd_vs_oracle_autograd(dev, preset, axes, ph, pw, box_warp):
    """d planes of the fused renderer vs autograd through the oracle's ImportanceRenderer.  'eg3d_original' takes the
    mirrored path (plane 2 = transpose of plane 1, not scattered), 'eg3d_fixed' the three-plane scatter."""
    import dataclasses
    from hfa_gp_amd import ops
    from hfa_gp_amd.config import PRESETS
    from hfa_gp_amd.generator import TriPlaneGenerator
    from oracle import eg3d_oracle as O
    cfg = dataclasses.replace(PRESETS[preset](), neural_rendering_resolution=10, img_resolution=40, plane_axes=axes)
    if box_warp is not None:
        cfg = dataclasses.replace(cfg, box_warp=box_warp)
    gen = perturb_state(TriPlaneGenerator(cfg, seed=0))
    P = state_cpu(gen)
    gen = gen.to(dev)
    c = look_at_label(torch.tensor([1.3, 1.8]), torch.tensor([1.5, 1.7]))
    g = torch.Generator().manual_seed(4)
    b, res = 2, cfg.neural_rendering_resolution
    r = res * res
    planes = torch.randn(b, 3, 32, ph, pw, generator=g, requires_grad=True)
    us = torch.rand(b, r, cfg.depth_resolution, 1, generator=g)
    ui = torch.rand(b * r, cfg.depth_resolution_importance, generator=g)
    g_feat = torch.randn(b, r, 32, generator=g)
    o, d = O.ray_sampler(c[:, :16].reshape(-1, 4, 4), c[:, 16:].reshape(-1, 3, 3), res)
    feat, _, _ = O.importance_renderer(P, cfg, planes, o, d, us, ui)
    feat.backward(g_feat)
    pl = planes.detach().permute(0, 1, 3, 4, 2).contiguous().to(dev)
    u_s, u_i = gen._uniforms(b, dev, us.to(dev), ui.to(dev))
    dpl = ops.raymarch_bwd(g_feat.to(dev), pl, u_strat=u_s, u_imp=u_i, **gen._render_args(c.to(dev)))
    close(dpl.permute(0, 1, 4, 2, 3), planes.grad, atol=2e-5, rtol=1e-3)


@pytest.mark.parametrize("preset,batch,prec", [("tiny64", 2, "fp32"), ("tiny14", 1, "fp32"), ("small128", 2, "fp32"),
                                               ("small128", 2, "bf16x3"), ("small128", 1, "bf16x6"),
                                               ("small128", 2, "f16x3")])
def test_synthesis_backward_vs_oracle_autograd(dev, preset, batch, prec):
    """dL/d ws for L = <image, G> + <image_raw, G_raw>, generator frozen (BASELINE config 3 mechanics)."""
    from hfa_gp_amd.config import PRESETS
    from hfa_gp_amd.generator import TriPlaneGenerator
    from oracle import eg3d_oracle as O
    cfg = dataclasses.replace(PRESETS[preset](), conv_precision=prec)
    k = {"fp32": 1.0, "bf16x6": 1.0, "bf16x3": 5.0, "f16x3": 5.0}[prec]     # (f16x3: gradient GEMMs run in bf16x3)
    gen = perturb_state(TriPlaneGenerator(cfg, seed=0)).requires_grad_(False)
    P = state_cpu(gen)
    gen = gen.to(dev)
    ws, c, us, ui = make_inputs(cfg, batch)
    g = torch.Generator().manual_seed(6)
    G = torch.randn(batch, 3, cfg.img_resolution, cfg.img_resolution, generator=g) / cfg.img_resolution
    G_raw = torch.randn(batch, 3, cfg.neural_rendering_resolution, cfg.neural_rendering_resolution, generator=g) / 64
    ws_ref = ws.clone().requires_grad_(True)
    ref = O.synthesis(P, cfg, ws_ref, c, us, ui)
    ((ref["image"] * G).sum() + (ref["image_raw"] * G_raw).sum()).backward()
    ws_d = ws.to(dev).requires_grad_(True)
    out = gen.synthesis(ws_d, c.to(dev), noise_mode="const", u_strat=us.to(dev), u_imp=ui.to(dev))
    close(out["image"], ref["image"], atol=1e-4 * k)
    ((out["image"] * G.to(dev)).sum() + (out["image_raw"] * G_raw.to(dev)).sum()).backward()
    scale = ws_ref.grad.abs().max().item()
    close(ws_d.grad, ws_ref.grad, atol=2e-4 * k * scale, rtol=2e-3 * k)


def test_trainer_step_on_gpu_matches_oracle_step(dev):
    """One `gen_update` (3DMM-driven, L2 only) on the MI355X vs the same step through the oracle on CPU:
    same loss, same gradients of bases / delta / driver net."""
    from hfa_gp_amd import headnerf
    from hfa_gp_amd.trainer import Trainer
    from tests.test_trainer_cpu import Args, OracleGenerator, frame

    def build(device, oracle):
        torch.manual_seed(0)
        gen = headnerf.HeadNeRF_3DMM(Args(), Args.size, device, 512, Args.latent_dim_shape)
        if oracle:
            OracleGenerator.adopt(gen.generator)
        tr = Trainer(Args(), device, mode="3dmm", gen=gen, lpips="none")
        tr.optimizer = torch.optim.SGD(tr.gen.parameters(), lr=0.0)    # compare gradients, not Adam's first step
        return tr

    real, label, params = frame(2)
    cpu = build("cpu", True)
    _, l2_cpu, _, _ = cpu.gen_update(real, label.clone(), params)
    gpu = build(dev, False)
    # same renderer uniforms as OracleGenerator draws (seed 0)
    cfg = gpu.gen.generator.cfg
    g = torch.Generator().manual_seed(0)
    r = cfg.neural_rendering_resolution ** 2
    us = torch.rand(1, r, cfg.depth_resolution, 1, generator=g).to(dev)
    ui = torch.rand(r, cfg.depth_resolution_importance, generator=g).to(dev)
    inner = gpu.gen.generator.synthesis
    gpu.gen.generator.synthesis = lambda ws, c=None, noise_mode="const": inner(ws, c, noise_mode, u_strat=us, u_imp=ui)
    _, l2_gpu, _, _ = gpu.gen_update(real.to(dev), label.clone().to(dev), params.to(dev))
    assert abs(float(l2_gpu) - float(l2_cpu)) < 1e-5
    for name in ("bases", "delta"):
        a, bref = getattr(gpu.gen, name).grad, getattr(cpu.gen, name).grad
        close(a, bref, atol=2e-4 * bref.abs().max().item(), rtol=2e-3)
    a = gpu.gen.weights_3dmm.fc[0].weight.grad
    bref = cpu.gen.weights_3dmm.fc[0].weight.grad
    close(a, bref, atol=2e-4 * bref.abs().max().item(), rtol=2e-3)


def test_rgb_trainer_and_render_harness_on_gpu(dev):
    """RGB-driven variant (Encoder on PyTorch-ROCm -> basis -> HIP generator): one fitting step moves the
    shared basis and the encoder, keeps the generator frozen; the batched reenactment harness returns uint8
    frames on the host with the reference's quantisation."""
    from hfa_gp_amd import headnerf
    from hfa_gp_amd.render import render_frames, to_uint8
    from hfa_gp_amd.trainer import Trainer

    class A:
        out_pose = False; person_2 = False; params_len = 76; size = 64; batch_size = 2; lr = 1e-3
        latent_dim_style = 512; latent_dim_shape = 8; generator_preset = "tiny14"; generator_seed = 0

    torch.manual_seed(0)
    tr = Trainer(A(), dev, mode="rgb", lpips="none")
    g0 = {k: v.clone() for k, v in tr.gen.generator.state_dict().items()}
    b0, e0 = tr.gen.bases.detach().clone(), tr.gen.encoder.fc[0].weight.detach().clone()
    g = torch.Generator().manual_seed(1)
    real = (0.5 * torch.randn(2, 3, 64, 64, generator=g)).clamp(-1, 1).to(dev)
    label = look_at_label(torch.tensor([1.5, 1.7]), torch.tensor([1.6, 1.5]), flipped=False).to(dev)
    l2a, _, img = tr.gen_update(real, label.clone())
    assert img.shape == (2, 3, 64, 64) and torch.isfinite(l2a)
    assert not torch.equal(tr.gen.bases.detach(), b0) and not torch.equal(tr.gen.encoder.fc[0].weight.detach(), e0)
    assert all(torch.equal(v, g0[k]) for k, v in tr.gen.generator.state_dict().items())
    for _ in range(5):
        l2b, _, _ = tr.gen_update(real, label.clone())
    assert float(l2b) < float(l2a)
    # after tune_generator() the same optimiser moves the generator too (trainer_rgb.py:58-60,69-71)
    tr.tune_generator()
    tr.gen_update(real, label.clone())
    moved = [k for k, v in tr.gen.generator.state_dict().items() if not torch.equal(v, g0[k])]
    assert any(k.endswith("b8.conv0.weight") for k in moved) and any(k.startswith("decoder.net.") for k in moved)
    assert any(k.endswith("torgb.affine.bias") for k in moved) and "backbone.synthesis.b4.const" in moved
    for p in tr.gen.generator.parameters():
        p.requires_grad_(False)
    # batched reenactment
    batches = [(real, label.clone()), (real.flip(0), label.clone())]
    frames = list(render_frames(tr.gen, batches))
    assert len(frames) == 2 and frames[0].dtype == torch.uint8 and frames[0].shape == (2, 3, 64, 64)
    assert not frames[0].is_cuda
    with torch.no_grad():
        tr.gen.eval()
        lab = label.clone()
        ref = to_uint8(tr.gen.get_image(tr.gen.get_latent(tr.gen.get_weights(real)), lab))
    # same frame up to the renderer's fresh uniforms: compare statistics, not pixels
    assert abs(frames[0].float().mean().item() - ref.float().mean().item()) < 8.0


@pytest.mark.parametrize("b,c,h,f", [(2, 3, 256, 2), (1, 3, 32, 2), (3, 3, 17, 1), (2, 3, 16, 4), (1, 1, 1, 2)])
def test_pool_mse_fused_loss(dev, b, c, h, f):
    """hfagp_pool_mse_fwd / _bwd vs AdaptiveAvgPool2d + MSELoss(mean) and their autograd (trainer_rgb.py:84-86)."""
    from hfa_gp_amd import ops
    g = torch.Generator().manual_seed(b * 10 + h)
    img = torch.randn(b, c, h * f, h * f, generator=g, requires_grad=True)
    real = (0.5 * torch.randn(b, c, h, h, generator=g)).clamp(-1, 1)
    pooled_ref = torch.nn.AdaptiveAvgPool2d((h, h))(img)
    loss_ref = F.mse_loss(real, pooled_ref, reduction="mean")
    (loss_ref * 1.7).backward()
    img_d = img.detach().to(dev).requires_grad_(True)
    loss, pooled = ops.pool_mse(img_d, real.to(dev))
    assert not pooled.requires_grad
    close(pooled, pooled_ref.detach(), atol=1e-6)
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 1e-6 + 1e-5 * abs(float(loss_ref.detach()))
    (loss * 1.7).backward()
    close(img_d.grad, img.grad, atol=1e-9 + 1e-5 * float(img.grad.abs().max()))
    loss2, _ = ops.pool_mse(img_d.detach(), real.to(dev))
    assert torch.equal(loss2, loss.detach()), "fixed reduction order -> bitwise repeatable"


def test_audio_trainer_and_batched_reenactment_on_gpu(dev):
    """Audio-driven variant (trainer_audio.py; BASELINE config 5 mechanics): AudioNet (+ AudioAttNet over the
    smoothing window) -> basis -> HIP generator.  One step per branch moves the right optimisers' parameters; the
    batched reenactment path renders N frames in one synthesis and equals the per-frame loop given the same
    renderer uniforms."""
    from hfa_gp_amd.trainer import AudioTrainer
    from tests.test_trainer_cpu import AudioArgs

    class A(AudioArgs):
        size = 64

    torch.manual_seed(0)
    auds = torch.randn(24, 16, 29, generator=torch.Generator().manual_seed(50)).numpy()
    tr = AudioTrainer(auds, 20, A(), dev, lpips="none")
    g = torch.Generator().manual_seed(1)
    real = (0.5 * torch.randn(1, 3, 64, 64, generator=g)).clamp(-1, 1).to(dev)
    label = look_at_label(torch.tensor([1.5]), torch.tensor([1.6]), flipped=False).to(dev)
    b0 = tr.gen.bases.detach().clone()
    a0 = next(tr.AudNet.parameters()).detach().clone()
    t0 = next(tr.AudAttNet.parameters()).detach().clone()
    _, l2, _, img = tr.gen_update(real, label.clone(), None, 0, torch.tensor([3], device=dev))
    assert torch.isfinite(l2) and img.shape == (1, 3, 64, 64)
    assert not torch.equal(tr.gen.bases.detach(), b0) and not torch.equal(next(tr.AudNet.parameters()).detach(), a0)
    assert torch.equal(next(tr.AudAttNet.parameters()).detach(), t0)
    tr.gen_update(real, label.clone(), None, A.nosmo_iters, 1)
    assert not torch.equal(next(tr.AudAttNet.parameters()).detach(), t0)
    # batched reenactment == per-frame loop when the renderer sees the same uniforms
    cfg = tr.gen.generator.cfg
    r = cfg.neural_rendering_resolution ** 2
    n = 5
    us = torch.rand(n, r, cfg.depth_resolution, device=dev)
    ui = torch.rand(n * r, cfg.depth_resolution_importance, device=dev)
    inner = tr.gen.generator.synthesis
    idx = torch.tensor([0, 2, 11, 22, 23], device=dev)
    tr.gen.generator.synthesis = lambda ws, c=None, noise_mode="const": inner(ws, c, noise_mode, u_strat=us, u_imp=ui)
    batched = tr.sample_frames(idx, label.repeat(n, 1))
    assert batched.shape == (n, 3, 64, 64)
    for k, i in enumerate(idx.tolist()):
        tr.gen.generator.synthesis = lambda ws, c=None, noise_mode="const", k=k: inner(
            ws, c, noise_mode, u_strat=us[k:k + 1], u_imp=ui[k * r:(k + 1) * r])
        one = tr.sample(None, label.clone(), None, A.nosmo_iters, i)
        close(batched[k:k + 1], one, atol=2e-4)


@pytest.mark.parametrize("b,h,cin,cout,mode", [(1, 4, 8, 32, "3x3"), (2, 9, 24, 96, "3x3"), (2, 21, 64, 128, "3x3"),
                                               (1, 4, 8, 32, "up"), (2, 11, 32, 64, "up"), (2, 13, 40, 96, "1x1")])
def test_conv_weight_gradient(dev, b, h, cin, cout, mode):
    """hfagp_conv_wgrad == autograd of the (modulated) convolution w.r.t. its weight (no demodulation term)."""
    from hfa_gp_amd import ops
    from oracle import eg3d_oracle as O
    g = torch.Generator().manual_seed(7)
    k = 1 if mode == "1x1" else 3
    x = torch.randn(b, cin, h, h, generator=g)
    s = torch.randn(b, cin, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g, requires_grad=True)
    xs = x * s[:, :, None, None]
    if mode == "up":
        y = O._conv_up2(xs, w, O.fir_kernel())
    else:
        y = F.conv2d(xs, w, padding=k // 2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xh = ops.nchw_to_nhwc(x.to(dev))
    gh = ops.nchw_to_nhwc(gy.to(dev))
    if mode == "up":
        dw = ops.conv_wgrad(xh, s.to(dev), ops.upfir_bwd(gh), w.detach().to(dev), ops.CONVT3X3_UP2)
    else:
        dw = ops.conv_wgrad(xh, s.to(dev), gh, w.detach().to(dev), ops.CONV3X3 if k == 3 else ops.CONV1X1)
    close(dw, w.grad, atol=1e-4 * w.grad.abs().max().item(), rtol=1e-4)


@pytest.mark.parametrize("b,h,w_,cin,cout", [(2, 21, 21, 64, 128), (1, 8, 40, 128, 64), (3, 5, 3, 64, 64), (1, 33, 16, 192, 128)])
def test_conv_weight_gradient_split_bf16(dev, b, h, w_, cin, cout):
    """The 3x3 weight gradient on the split-bf16 MFMA kernel (wgrad_bf16.hip; ragged position tiles, several channel
    tiles, several split-K slabs) vs autograd; products carry 2^-16, sums are fp32."""
    from hfa_gp_amd import ops
    g = torch.Generator().manual_seed(17)
    x = torch.randn(b, cin, h, w_, generator=g)
    s = torch.randn(b, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g, requires_grad=True)
    y = F.conv2d(x * s[:, :, None, None], w, padding=1)
    gy = torch.randn(y.shape, generator=g) * 1e-6            # gradients are small numbers: bf16 parts keep the exponent
    y.backward(gy)
    xh, gh = ops.nchw_to_nhwc(x.to(dev)), ops.nchw_to_nhwc(gy.to(dev))
    dw = ops.conv_wgrad(xh, s.to(dev), gh, w.detach().to(dev), ops.CONV3X3, precision="bf16x3")
    exact = ops.conv_wgrad(xh, s.to(dev), gh, w.detach().to(dev), ops.CONV3X3)
    scale = w.grad.abs().max().item()
    close(exact, w.grad, atol=1e-4 * scale, rtol=1e-4)
    close(dw, w.grad, atol=1e-4 * scale, rtol=1e-4)
    assert not torch.equal(dw, exact), "the split kernel did not run"


@pytest.mark.parametrize("b,h,w_,cin,cout", [(2, 11, 11, 64, 128), (1, 8, 21, 128, 64), (3, 3, 5, 64, 64), (1, 32, 16, 128, 128),
                                             (2, 20, 33, 32, 128), (1, 15, 16, 192, 64)])
def test_upconv_weight_gradient_split_bf16(dev, b, h, w_, cin, cout):
    """The weight gradient of the up-sampling conv on the split-bf16 MFMA kernel (round 5: ONE launch, x staged once per position
    tile of the parity grid, four parity phases into the nine accumulators; Cin = 32 — the first super-resolution layer — as a
    partial channel tile) vs autograd through the oracle's transposed conv + FIR; ragged tiles, W + 1 a multiple of 16 + 1."""
    from hfa_gp_amd import ops
    from oracle import eg3d_oracle as O
    g = torch.Generator().manual_seed(19)
    x = torch.randn(b, cin, h, w_, generator=g)
    s = torch.randn(b, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g, requires_grad=True)
    y = O._conv_up2(x * s[:, :, None, None], w, O.fir_kernel())
    gy = torch.randn(y.shape, generator=g) * 1e-5
    y.backward(gy)
    xh = ops.nchw_to_nhwc(x.to(dev))
    gph = ops.upfir_bwd(ops.nchw_to_nhwc(gy.to(dev)))
    dw = ops.conv_wgrad(xh, s.to(dev), gph, w.detach().to(dev), ops.CONVT3X3_UP2, precision="bf16x3")
    exact = ops.conv_wgrad(xh, s.to(dev), gph, w.detach().to(dev), ops.CONVT3X3_UP2)
    scale = w.grad.abs().max().item()
    close(exact, w.grad, atol=1e-4 * scale, rtol=1e-4)
    close(dw, w.grad, atol=1e-4 * scale, rtol=1e-4)
    assert not torch.equal(dw, exact), "the split kernel did not run"


@pytest.mark.parametrize("preset,batch,prec", [("tiny64", 2, "fp32"), ("tiny14", 1, "fp32"), ("small128", 1, "fp32"),
                                               ("small128", 1, "bf16x3")])
def test_generator_parameter_gradients_vs_oracle_autograd(dev, preset, batch, prec):
    """tune_generator() mode: dL/d(every generator parameter) and dL/d ws against autograd through the oracle.
    (The bwd-data GEMMs follow conv_precision — bf16x3 for the fp16 kinds; the 3x3 weight-gradient GEMMs run on split
    bf16 unless conv_precision is 'fp32'; the parity / 1x1 ones are always exact fp32.)"""
    from hfa_gp_amd.config import PRESETS
    from hfa_gp_amd.generator import TriPlaneGenerator
    from oracle import eg3d_oracle as O
    cfg = dataclasses.replace(PRESETS[preset](), conv_precision=prec)
    k = {"fp32": 1.0, "bf16x3": 5.0}[prec]
    gen = perturb_state(TriPlaneGenerator(cfg, seed=0))
    P = {k: v.detach().clone() for k, v in gen.state_dict().items()}
    names = [n for n, _ in gen.named_parameters() if not n.startswith("backbone.mapping.")]
    for n in names:
        P[n].requires_grad_(True)
    gen = gen.to(dev)
    ws, c, us, ui = make_inputs(cfg, batch)
    g = torch.Generator().manual_seed(8)
    G = torch.randn(batch, 3, cfg.img_resolution, cfg.img_resolution, generator=g) / cfg.img_resolution
    ws_ref = ws.clone().requires_grad_(True)
    (O.synthesis(P, cfg, ws_ref, c, us, ui)["image"] * G).sum().backward()
    # the CANCELLATION SCALE of every noise-strength gradient: that gradient is one number, sum_pixels g(pixel) *
    # noise(pixel), with near-total cancellation — |sum| ~ 1e-3 of sum|.| — so an error bar relative to |sum| says nothing.
    # A second oracle pass with each strength spread over its noise map as a per-pixel tensor of the same value returns the
    # per-pixel terms: their absolute sum is the scale the error of ANY summation order is proportional to.
    P2 = {k: v.detach().clone() for k, v in P.items()}
    spread = {}
    for n in names:
        if n.endswith(".noise_strength"):
            spread[n] = torch.full_like(P2[n[:-len("noise_strength")] + "noise_const"], float(P2[n])).requires_grad_(True)
            P2[n] = spread[n]
    (O.synthesis(P2, cfg, ws.clone(), c, us, ui)["image"] * G).sum().backward()
    cancel = {n: float(t.grad.abs().sum()) for n, t in spread.items() if t.grad is not None}
    for n, t in spread.items():
        if t.grad is not None and P[n].grad is not None:        # the spread pass reproduces the scalar gradient
            assert abs(float(t.grad.sum()) - float(P[n].grad)) <= 1e-3 * cancel[n] + 1e-9, n
    ws_d = ws.to(dev).requires_grad_(True)
    out = gen.synthesis(ws_d, c.to(dev), noise_mode="const", u_strat=us.to(dev), u_imp=ui.to(dev))
    (out["image"] * G.to(dev)).sum().backward()
    close(ws_d.grad, ws_ref.grad, atol=2e-4 * k * ws_ref.grad.abs().max().item(), rtol=2e-3 * k)
    params = dict(gen.named_parameters())
    bad = []
    for n in names:
        ref = P[n].grad
        got = params[n].grad
        if ref is None:                       # unused by the path (SR noise_strength with noise_mode 'none')
            assert got is None or float(got.abs().max()) == 0.0, n
            continue
        assert got is not None, n
        scale = max(ref.abs().max().item(), 1e-12)
        err = (got.cpu() - ref).abs().max().item()
        # (a noise_strength gradient is ONE number with near-total cancellation — sum |g * noise| is ~1e3 x |sum|.  Its bar is
        #  relative to that cancellation scale, but with a constant that still means something for the VALUE (5e-4 k of the
        #  cancellation scale was 50 % of the value in fp32 and passed a sign flip at bf16x3): 1e-4 k of the cancellation scale, i.e.
        #  ~10 % of the value at fp32 — the oracle's own fp32 run sits 2-3 % from the fp64 truth (three-way test) — and a sign flip,
        #  an error of 2 |sum| = 2e-3 of the scale, fails at every k.)
        tol = 5e-4 * k
        if n in cancel:
            scale = max(cancel[n], 1e-12)
            tol = 1e-4 * k
        if not err <= tol * scale + 1e-7:
            bad.append((n, err, scale))
    assert not bad, bad[:8]


@pytest.mark.parametrize("m,n", [(7168, 50), (7168, 8), (7168, 64), (512, 64)])
def test_tall_skinny_qr_matches_torch(dev, m, n):
    """ops.TallSkinnyQR (Gram-matrix Householder kernel) against torch.linalg.qr on the CPU: same Q
    including LAPACK's column signs, and the same gradient for a random upstream dQ (headnerf.py:85-98).  (The
    method squares the condition number: it is for tall, well-conditioned panels like the latent basis; HeadNeRF
    falls back to torch.linalg.qr for m < 8 n.)"""
    from hfa_gp_amd import ops
    g = torch.Generator().manual_seed(m + n)
    bases = torch.randn(n, m, generator=g)
    a_ref = (bases.double() + 1e-8).T.clone().requires_grad_(True)
    q_ref = torch.linalg.qr(a_ref, mode="reduced")[0]
    gq = torch.randn(m, n, generator=g)
    (q_ref * gq.double()).sum().backward()
    b_d = bases.to(dev).requires_grad_(True)
    q = ops.TallSkinnyQR.apply((b_d + 1e-8).T)
    (q * gq.to(dev)).sum().backward()
    close(q, q_ref.float(), atol=2e-6)
    eye = torch.eye(n)
    close((q.T @ q).cpu(), eye, atol=5e-6)
    scale = a_ref.grad.abs().max().item()
    close(b_d.grad.T, a_ref.grad.float(), atol=2e-5 * scale)


def test_get_latent_fast_qr_matches_torch_qr(dev):
    """The latent-basis map ws = alpha @ Q^T + delta with Q from TallSkinnyQR against the same map with
    torch.linalg.qr on the CPU: ws and the gradients w.r.t. bases, delta and alpha."""
    from hfa_gp_amd import ops
    g = torch.Generator().manual_seed(3)
    k, dim = 50, 512
    bases = torch.randn(k, 14 * dim, generator=g)
    delta = bases.mean(0)
    alpha = torch.randn(3, k, generator=g)
    gws = torch.randn(3, 14, dim, generator=g)

    def latent(b, d, al, qr):
        q = qr((b + 1e-8).T)
        return (al @ q.T).view(al.shape[0], -1, dim) + d.view(-1, dim)
    ref_in = [t.double().clone().requires_grad_(True) for t in (bases, delta, alpha)]
    ws_ref = latent(*ref_in, lambda a: torch.linalg.qr(a, mode="reduced")[0])
    (ws_ref * gws.double()).sum().backward()
    got_in = [t.to(dev).requires_grad_(True) for t in (bases, delta, alpha)]
    ws = latent(*got_in, ops.TallSkinnyQR.apply)
    (ws * gws.to(dev)).sum().backward()
    close(ws, ws_ref.float(), atol=2e-5)
    for got, ref in zip(got_in, ref_in):
        close(got.grad, ref.grad.float(), atol=2e-5 * ref.grad.abs().max().item() + 1e-7)

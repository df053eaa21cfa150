"""Pins for the CPU oracle (oracle/eg3d_oracle.py).

The generator itself is "parity unpinned" (EG3D is absent from the reference tree), so what CAN be
pinned is pinned here: (a) the FIR / bias-act / equalised-linear conventions against golden vectors
produced by the reference's own code (encoder3d.py `upfirdn2d_native`, `fused_leaky_relu`,
`EqualLinear`), (b) bilinear sampling / convolutions against ATen, (c) closed-form cases of the
compositing and sampling maths."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import eg3d_oracle as O
from tests.util import ROOT, look_at_label

G = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"), allow_pickle=False)


def T(name):
    return torch.from_numpy(G[name])


# ----------------------------------------------------------------------------- (a) reference-pinned
@pytest.mark.parametrize("name,up,down,pad", [
    ("u1d1p21", 1, 1, (2, 1)), ("u1d1p11", 1, 1, (1, 1)), ("u2d1p21", 2, 1, (2, 1)),
    ("u1d2p11", 1, 2, (1, 1)), ("u2d2p21", 2, 2, (2, 1)), ("u1d1p0m1", 1, 1, (0, -1))])
def test_upfirdn2d_matches_reference_native(name, up, down, pad):
    y = O.upfirdn2d(T("fir_x"), T("fir_k"), up=up, down=down, padding=(pad[0], pad[1], pad[0], pad[1]))
    assert torch.allclose(y, T("fir_" + name), atol=1e-6)


def test_upsample2d_matches_reference_native():
    assert torch.allclose(O.fir_kernel((1, 3, 3, 1)), T("fir_k"), atol=0)
    assert torch.allclose(O.upsample2d(T("fir_x"), O.fir_kernel()), T("fir_up2gain4"), atol=1e-6)


def test_bias_act_matches_reference_fused_leaky_relu():
    y = O.bias_act(T("fir_x"), T("flrelu_b").reshape(-1), act="lrelu")
    assert torch.allclose(y, T("flrelu_y"), atol=1e-6)


def test_fully_connected_matches_reference_equal_linear():
    y = O.fully_connected(T("eqlin_x"), T("eqlin_w"), T("eqlin_b"), lr_mul=0.5)
    assert torch.allclose(y, T("eqlin_y"), atol=1e-6)


# ----------------------------------------------------------------------------- (b) ATen cross-checks
def test_modconv_forms_agree_and_reduce_to_plain_conv():
    torch.manual_seed(0)
    x, w = torch.randn(2, 8, 9, 9), torch.randn(16, 8, 3, 3)
    s = torch.randn(2, 8)
    f = O.fir_kernel()
    for up in (1, 2):
        a = O.modulated_conv2d(x, w, s, up=up, f=f, fused=True)
        b = O.modulated_conv2d(x, w, s, up=up, f=f, fused=False)
        assert torch.allclose(a, b, atol=2e-5)
    ones = torch.ones(2, 8)
    y = O.modulated_conv2d(x, w, ones, demodulate=False)
    assert torch.allclose(y, F.conv2d(x, w, padding=1), atol=1e-5)


def test_up_conv_equals_zero_insert_true_convolution():
    """conv_transpose(stride 2) + FIR(pad 1, gain 4) == FIR(upsample) of a true convolution (§11.2)."""
    torch.manual_seed(1)
    x, w = torch.randn(1, 4, 8, 8), torch.randn(6, 4, 3, 3)
    f = O.fir_kernel()
    y = O._conv_up2(x, w, f)
    z = x.new_zeros(1, 4, 8, 2, 8, 2)
    z[:, :, :, 0, :, 0] = x
    z = z.reshape(1, 4, 16, 16)
    full = F.conv2d(F.pad(z, [2, 2, 2, 2]), w.flip([2, 3]))            # true convolution, 'full' size 18
    ref = O.upfirdn2d(full[:, :, :17, :17], f, padding=(1, 1, 1, 1), gain=4.0)
    assert y.shape == (1, 6, 16, 16)
    assert torch.allclose(y, ref, atol=2e-5)


def test_plane_projection_axes():
    pts = torch.tensor([[[0.1, 0.2, 0.3]]])
    planes = torch.zeros(1, 3, 1, 4, 4)
    for kind, third in (("eg3d_original", (0.3, 0.1)), ("eg3d_fixed", (0.3, 0.2))):
        axes = O.plane_axes(kind)
        proj = torch.bmm(pts.expand(3, -1, -1), torch.linalg.inv(axes))[..., :2]
        assert torch.allclose(proj[0, 0], torch.tensor([0.1, 0.2]))
        assert torch.allclose(proj[1, 0], torch.tensor([0.1, 0.3]))
        assert torch.allclose(proj[2, 0], torch.tensor(third))
    assert O.sample_from_planes(O.plane_axes("eg3d_original"), planes, pts, 1.0).shape == (1, 3, 1, 1)


def test_ray_sampler_pinhole_corners():
    """Frontal camera at (0,0,2.7): pixel (row i, col j) looks through ((j+.5)/res-.5)/fx, -((i+.5)/res-.5)/fy."""
    c = look_at_label(torch.tensor([math.pi / 2]), torch.tensor([math.pi / 2]))
    c2w, intr = c[:, :16].reshape(1, 4, 4), c[:, 16:].reshape(1, 3, 3)
    res = 8
    o, d = O.ray_sampler(c2w, intr, res)
    assert torch.allclose(o[0, 0], torch.tensor([0.0, 0.0, 2.7]), atol=1e-6)
    for (i, j) in [(0, 0), (0, res - 1), (res - 1, 0), (res - 1, res - 1), (3, 5)]:
        xc, yc = ((j + 0.5) / res - 0.5) / 4.2647, ((i + 0.5) / res - 0.5) / 4.2647
        want = F.normalize(torch.tensor([xc, -yc, -1.0]), dim=0)    # camera looks down -z, image y flips
        assert torch.allclose(d[0, i * res + j], want, atol=1e-5), (i, j)


# ----------------------------------------------------------------------------- (c) closed forms
def test_ray_march_constant_medium_closed_form():
    s, sigma, delta, col = 12, 0.7, 0.05, 0.3
    depths = (2.0 + delta * torch.arange(s, dtype=torch.float32)).reshape(1, 1, s, 1)
    dens = torch.full((1, 1, s, 1), sigma)
    colors = torch.full((1, 1, s, 4), col)
    rgb, depth, w = O.ray_march(colors, dens, depths)
    st = math.log1p(math.exp(sigma - 1.0))
    i = torch.arange(s - 1, dtype=torch.float64)
    want_w = torch.exp(-st * delta * i) * (1 - math.exp(-st * delta))
    assert torch.allclose(w.reshape(-1).double(), want_w, atol=1e-6)
    total = 1 - math.exp(-st * delta * (s - 1))
    assert torch.allclose(rgb, torch.full_like(rgb, col * total * 2 - 1), atol=1e-6)
    t_mid = 2.0 + delta * (i + 0.5)
    assert abs(depth.item() - float((want_w * t_mid).sum() / want_w.sum())) < 1e-5


def test_ray_march_empty_space_depth_is_clamped_not_nan():
    s = 6
    depths = torch.linspace(2.25, 3.3, s).reshape(1, 1, s, 1)
    dens = torch.full((1, 1, s, 1), -1e4)                      # softplus -> exactly 0 weight
    rgb, depth, w = O.ray_march(torch.rand(1, 1, s, 3), dens, depths)
    assert float(w.sum()) == 0.0 and torch.isfinite(depth).all()
    assert abs(depth.item() - 3.3) < 1e-6 and torch.allclose(rgb, -torch.ones_like(rgb))


def test_importance_sampling_of_uniform_weights_is_linear():
    s = 16
    z = torch.linspace(2.25, 3.3, s).reshape(1, 1, s, 1)
    w = torch.full((1, 1, s - 1, 1), 0.02)
    u = torch.linspace(0.0, 0.999, 7).reshape(1, 7)
    t = O.sample_importance(z, w, u).reshape(-1)
    mids = 0.5 * (z.reshape(-1)[:-1] + z.reshape(-1)[1:])
    want = mids[0] + u.reshape(-1) * (mids[s - 3] - mids[0])   # pdf cells span z_mid[0] .. z_mid[S-3]
    assert torch.allclose(t, want, atol=1e-5)


def test_zero_planes_give_decoder_bias_everywhere():
    from hfa_gp_amd.config import tiny64
    from hfa_gp_amd.generator import TriPlaneGenerator
    from tests.util import make_inputs, perturb_state, state_cpu
    cfg = tiny64()
    P = state_cpu(perturb_state(TriPlaneGenerator(cfg)))
    ws, c, us, ui = make_inputs(cfg, 1)
    o, d = O.ray_sampler(c[:, :16].reshape(1, 4, 4), c[:, 16:].reshape(1, 3, 3), 4)
    planes = torch.zeros(1, 3, 32, 8, 8)
    feat, depth, wsum = O.importance_renderer(P, cfg, planes, o, d, us[:, :16], ui[:16])
    hid = F.softplus(P["decoder.net.0.bias"])
    out = hid @ (P["decoder.net.2.weight"] / 8.0).t() + P["decoder.net.2.bias"]
    col = torch.sigmoid(out[1:]) * 1.002 - 0.001
    want = col * wsum[0, 0] * 2 - 1                              # constant colour along every ray
    assert torch.allclose(feat[0, 0], want, atol=1e-5)


def test_synthesis_is_per_sample_independent():
    from hfa_gp_amd.config import tiny64
    from hfa_gp_amd.generator import TriPlaneGenerator
    from tests.util import make_inputs, state_cpu
    cfg = tiny64()
    P = state_cpu(TriPlaneGenerator(cfg))
    ws, c, us, ui = make_inputs(cfg, 2)
    r = cfg.neural_rendering_resolution ** 2
    both = O.synthesis(P, cfg, ws, c, us, ui)["image"]
    one = O.synthesis(P, cfg, ws[1:], c[1:], us[1:], ui[r:])["image"]
    assert torch.allclose(both[1:], one, atol=1e-5)

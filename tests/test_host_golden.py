"""Host-side mirror of the reference's module API (cam_utils, encoder3d, headnerf) against golden vectors
captured from the reference itself (tests/golden/make_golden.py)."""
import math
import os

import numpy as np
import pytest
import torch

from hfa_gp_amd import cam_utils, encoder3d, headnerf
from tests.util import ROOT

G = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"), allow_pickle=False)


def T(name):
    return torch.from_numpy(G[name])


class Args:
    out_pose = False
    person_2 = False
    params_len = 76
    generator_preset = "tiny64"      # keep the CPU-side construction small; the basis layer ignores it


class StubGenerator(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.calls = []

    def synthesis(self, ws, c=None, noise_mode=None):
        self.calls.append((ws.detach().clone(), c.detach().clone(), noise_mode))
        return {"image": ws.mean(dim=(1, 2)).view(-1, 1, 1, 1).expand(-1, 3, 4, 4)}


# ----------------------------------------------------------------------------- cam_utils
def test_cam2world_fixed_angles():
    for i, (h, v) in enumerate(G["cam_hv"]):
        p, phi, theta = cam_utils.sample_camera_positions("cpu", n=1, r=2.7, horizontal_mean=float(h),
                                                          vertical_mean=float(v), mode=None)
        assert torch.allclose(p, T("cam_points")[i: i + 1], atol=1e-6)
        m = cam_utils.create_cam2world_matrix(-p, p, device="cpu")
        assert torch.allclose(m, T("cam_c2w")[i: i + 1], atol=1e-6)


def test_frontal_camera_is_identity_rotation_at_radius():
    p, _, _ = cam_utils.sample_camera_positions("cpu", n=1, r=2.7, mode=None)
    m = cam_utils.create_cam2world_matrix(-p, p, device="cpu")[0]
    want = torch.eye(4)
    want[2, 3] = 2.7
    assert torch.allclose(m, want, atol=1e-6)


def test_gaussian_sampling_consumes_rng_like_reference():
    torch.manual_seed(20)
    p, phi, theta = cam_utils.sample_camera_positions("cpu", n=6, r=2.7, horizontal_stddev=0.3,
                                                      vertical_stddev=0.155, mode="gaussian")
    assert torch.allclose(p, T("cam_gauss_points"), atol=1e-6)
    assert torch.allclose(phi, T("cam_gauss_phi"), atol=1e-6) and torch.allclose(theta, T("cam_gauss_theta"), atol=1e-6)
    assert torch.allclose(cam_utils.create_cam2world_matrix(-p, p, device="cpu"), T("cam_gauss_c2w"), atol=1e-6)
    torch.manual_seed(20)
    lab = cam_utils.cam_sampler(6, "cpu")
    assert lab.shape == (6, 25) and torch.allclose(lab[:, :16], T("cam_gauss_c2w").reshape(6, 16), atol=1e-6)
    assert torch.allclose(lab[0, 16:], torch.tensor(cam_utils.FFHQ_INTRINSICS))


# ----------------------------------------------------------------------------- encoder3d
@pytest.mark.parametrize("name,up,down,pad", [
    ("u1d1p21", 1, 1, (2, 1)), ("u1d1p11", 1, 1, (1, 1)), ("u2d1p21", 2, 1, (2, 1)),
    ("u1d2p11", 1, 2, (1, 1)), ("u2d2p21", 2, 2, (2, 1)), ("u1d1p0m1", 1, 1, (0, -1))])
def test_upfirdn2d(name, up, down, pad):
    y = encoder3d.upfirdn2d(T("fir_x"), T("fir_k"), up=up, down=down, pad=pad)
    assert torch.allclose(y, T("fir_" + name), atol=1e-6)


def test_small_layers():
    assert torch.allclose(encoder3d.make_kernel([1, 3, 3, 1]), T("fir_k"))
    assert torch.allclose(encoder3d.fused_leaky_relu(T("fir_x"), T("flrelu_b")), T("flrelu_y"), atol=1e-6)
    lin = encoder3d.EqualLinear(16, 8, lr_mul=0.5, bias_init=0.3)
    with torch.no_grad():
        lin.weight.copy_(T("eqlin_w")); lin.bias.copy_(T("eqlin_b"))
        assert torch.allclose(lin(T("eqlin_x")), T("eqlin_y"), atol=1e-6)
        conv = encoder3d.EqualConv2d(4, 6, 3, stride=1, padding=1)
        conv.weight.copy_(T("eqconv_w")); conv.bias.copy_(T("eqconv_b"))
        assert torch.allclose(conv(T("fir_x")), T("eqconv_y"), atol=1e-5)


def test_resblock_state_dict_and_output():
    rb = encoder3d.ResBlock(8, 16)
    sd = {k[len("resblock_sd/"):]: T(k) for k in G.files if k.startswith("resblock_sd/")}
    assert sorted(sd) == sorted(rb.state_dict())
    rb.load_state_dict(sd, strict=True)
    with torch.no_grad():
        assert torch.allclose(rb(T("resblock_x")), T("resblock_y"), atol=1e-5)


def test_encoder_keys_seeded_init_and_output():
    torch.manual_seed(int(G["enc64_seed"][0]))
    enc = encoder3d.Encoder(64, 512, 50, False, False)
    sd = enc.state_dict()
    assert list(sd.keys()) == [str(k) for k in G["enc64_keys"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in G["enc64_shapes"]]
    assert torch.allclose(sd["net_app.convs.1.conv1.0.weight"].flatten()[:16], T("enc64_w_probe"))
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(int(G["enc64_seed"][1])))
    with torch.no_grad():
        y = enc(x)
    assert torch.allclose(y, T("enc64_y"), atol=2e-4, rtol=1e-4)


# ----------------------------------------------------------------------------- driver nets
def test_weights_3dmm():
    torch.manual_seed(6)
    m = headnerf.Weights_3DMM(76, 512, 50)
    assert list(m.state_dict()) == [str(k) for k in G["w3dmm_keys"]]
    x = torch.randn(3, 76, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        assert torch.allclose(m(x), T("w3dmm_y"), atol=1e-4, rtol=1e-4)


def test_audio_nets():
    torch.manual_seed(8)
    an = headnerf.AudioNet(64, 16)
    assert list(an.state_dict()) == [str(k) for k in G["audnet_keys"]]
    x = torch.randn(8, 16, 29, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        assert torch.allclose(an(x), T("audnet_y"), atol=1e-5)
    torch.manual_seed(10)
    aa = headnerf.AudioAttNet()
    assert list(aa.state_dict()) == [str(k) for k in G["audatt_keys"]]
    x = torch.randn(8, 64, generator=torch.Generator().manual_seed(11))
    with torch.no_grad():
        assert torch.allclose(aa(x), T("audatt_y"), atol=1e-5)


# ----------------------------------------------------------------------------- latent-basis layer
@pytest.mark.parametrize("K", [8, 50])
def test_get_latent_forward_backward(K):
    torch.manual_seed(12)
    m = headnerf.HeadNeRF_3DMM(Args(), 64, "cpu", 512, K)
    keys = [k for k in m.state_dict() if not k.startswith("generator.")]
    assert keys == [str(k) for k in G[f"hn{K}_keys"] if not str(k).startswith("generator.")]
    assert any(k.startswith("generator.backbone.synthesis.b4.") for k in m.state_dict())
    if K == 8:
        assert torch.allclose(m.bases.detach(), T("hn8_bases")) and torch.allclose(m.delta.detach(), T("hn8_delta"))
    else:
        assert torch.allclose(m.bases.detach()[:, :8], T("hn50_bases_probe"))
    alpha = T(f"hn{K}_alpha").clone().requires_grad_(True)
    ws = m.get_latent(alpha)
    assert ws.shape == (2, 14, 512)
    got = ws.detach() if K == 8 else ws.detach()[:, :, :16]
    assert torch.allclose(got, T(f"hn{K}_ws"), atol=2e-5)
    up = torch.randn(ws.shape, generator=torch.Generator().manual_seed(14))
    (ws * up).sum().backward()
    assert torch.allclose(alpha.grad, T(f"hn{K}_dalpha"), atol=2e-4, rtol=1e-4)
    dd = m.delta.grad if K == 8 else m.delta.grad[:64]
    db = m.bases.grad if K == 8 else m.bases.grad[:, :64]
    assert torch.allclose(dd, T(f"hn{K}_ddelta"), atol=1e-6)
    assert torch.allclose(db, T(f"hn{K}_dbases"), atol=2e-4, rtol=1e-3)
    # frozen basis → cached orthonormal factor, identical values
    m.bases.requires_grad_(False)
    with torch.no_grad():
        a = m.get_latent(alpha.detach())
        q1 = m._q_cache[1]
        b = m.get_latent(alpha.detach())
        assert m._q_cache[1] is q1 and torch.equal(a, b)
        assert torch.allclose(a, ws.detach(), atol=1e-6)
        assert float((q1.T @ q1 - torch.eye(K)).abs().max()) < 1e-5


def test_label_flip_in_place_and_alternation():
    torch.manual_seed(15)
    m = headnerf.HeadNeRF_3DMM(Args(), 64, "cpu", 512, 8)
    m.generator = StubGenerator()
    label = T("flip_label_before").clone()
    m.get_image(torch.ones(2, 14, 512), label)
    assert torch.equal(label, T("flip_label_after1"))                      # caller's tensor mutated
    assert torch.equal(m.generator.calls[-1][1], T("flip_seen_by_generator1"))
    assert m.generator.calls[-1][2] == str(G["flip_noise_mode"])
    m.get_image(torch.ones(2, 14, 512), label)
    assert torch.equal(label, T("flip_label_after2")) and torch.equal(label, T("flip_label_before"))
    label3 = label.clone()
    m(T("fwd_params"), label3)
    assert torch.equal(label3, T("fwd_label_after"))
    assert torch.allclose(m.generator.calls[-1][0][:, :, :8], T("fwd_ws_seen"), atol=2e-5)


def test_layout_grid_quantisation_formula():
    """(img * 127.5 + 128).clamp(0, 255).uint8 on a [-1.2, 1.2] ramp, tiled 1 x 2 (run_recon_video_rgb.py:28-42)."""
    from hfa_gp_amd.render import layout_grid
    out = layout_grid(T("grid_in"), grid_w=2, grid_h=1)
    assert out.dtype == np.uint8 and np.array_equal(out, G["grid_out"])


def test_converter_key_selection_round_trip(tmp_path):
    """tools/convert_eg3d_pickle.py: an EG3D-style state dict (extra keys, same names) and an HFA-GP checkpoint
    (`generator.` prefix) both map onto the generator's state dict; shape mismatches are refused."""
    import importlib.util
    import pytest
    import torch
    from safetensors.torch import load_file, save_file
    from hfa_gp_amd.config import tiny64
    from hfa_gp_amd.generator import TriPlaneGenerator
    spec = importlib.util.spec_from_file_location("convert_eg3d_pickle", os.path.join(ROOT, "tools", "convert_eg3d_pickle.py"))
    conv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(conv)
    src_gen = TriPlaneGenerator(tiny64(), seed=11)
    sd = {k: v.clone() for k, v in src_gen.state_dict().items()}
    sd["rendering_kwargs_placeholder"] = torch.zeros(1)                       # keys EG3D has and we do not
    picked = conv.select_for(TriPlaneGenerator(tiny64(), seed=0), sd)
    assert set(picked) == set(src_gen.state_dict())
    path = str(tmp_path / "g.safetensors")
    save_file(picked, path)
    dst = TriPlaneGenerator(tiny64(), seed=0)
    dst.load_state_dict(load_file(path), strict=True)
    for k, v in src_gen.state_dict().items():
        assert torch.equal(dst.state_dict()[k], v), k
    ck = str(tmp_path / "000100.pt")
    torch.save({"gen": {"generator." + k: v for k, v in sd.items()} | {"bases": torch.zeros(2, 2)}}, ck)
    assert set(conv.hfagp_state_dict(ck)) == set(sd)
    bad = dict(sd)
    bad["decoder.net.0.weight"] = torch.zeros(3, 3)
    with pytest.raises(SystemExit, match="shape mismatch"):
        conv.select_for(dst, bad)


def _eg3d_keys():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "eg3d_ffhq512_128_keys.json")))["keys"]


def test_generator_state_dict_matches_eg3d_key_fixture():
    """The ffhq512_128 generator registers exactly the names and shapes of EG3D's `G_ema.state_dict()` for
    ffhqrebalanced512-128 (fixture enumerated independently of the package, tests/golden/make_eg3d_keys.py; provenance:
    recalled structure, not the real pickle)."""
    import torch
    from hfa_gp_amd.config import ffhq512_128
    from hfa_gp_amd.generator import TriPlaneGenerator
    want = _eg3d_keys()
    with torch.device("meta"):
        have = {k: list(v.shape) for k, v in TriPlaneGenerator(ffhq512_128()).state_dict().items()}
    assert set(have) == set(want), (sorted(set(have) - set(want))[:5], sorted(set(want) - set(have))[:5])
    assert all(have[k] == want[k] for k in want), [k for k in want if have[k] != want[k]][:5]
    assert len(want) == 176


def test_strict_load_of_an_hfagp_checkpoint_built_from_the_eg3d_key_set(tmp_path):
    """trainer_rgb.py:130-151: `self.gen.module.load_state_dict(ckpt["gen"])` — STRICT — where ckpt["gen"] is the
    HeadNeRF_final state dict: bases, delta, encoder.*, and every EG3D tensor under `generator.` (plus, depending on the
    EG3D version, helper buffers of modules that are fused away here).  A synthetic checkpoint with those keys must load
    strictly into this package's HeadNeRF_final, through Trainer.resume, and through the offline converter."""
    import importlib.util
    import torch
    from hfa_gp_amd import headnerf
    from hfa_gp_amd.trainer import Trainer

    class A:
        out_pose = False; person_2 = False; params_len = 76; size = 256; batch_size = 1; lr = 3e-4
        latent_dim_style = 512; latent_dim_shape = 50; generator_preset = "ffhq512_128"; generator_seed = 0

    torch.manual_seed(0)
    model = headnerf.HeadNeRF_final(A(), 256, "cpu", 512, 50)
    g = torch.Generator().manual_seed(1)
    gen_sd = {"generator." + k: torch.randn(shape, generator=g) if shape else torch.randn((), generator=g)
              for k, shape in _eg3d_keys().items()}
    gen_sd["generator.renderer.plane_axes"] = torch.zeros(3, 3, 3)              # version-dependent helper buffers
    gen_sd["generator.superresolution.resample_filter"] = torch.zeros(4, 4)
    sd = {k: v.clone() for k, v in model.state_dict().items() if not k.startswith("generator.")}
    assert {"bases", "delta"} <= set(sd) and any(k.startswith("encoder.net_app.convs.") for k in sd)
    sd.update(gen_sd)
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert sorted(model.generator.ignored_checkpoint_keys) == ["renderer.plane_axes", "superresolution.resample_filter"]
    assert torch.equal(model.generator.decoder.net["2"].weight, gen_sd["generator.decoder.net.2.weight"])
    # a key that is NOT a known helper buffer still fails the strict load
    bad = dict(sd)
    bad["generator.backbone.synthesis.b1024.conv0.weight"] = torch.zeros(1)
    import pytest
    with pytest.raises(RuntimeError, match="b1024"):
        model.load_state_dict(bad, strict=True)
    # Trainer.resume on a reference-format checkpoint file
    tr = Trainer(A(), "cpu", mode="rgb", lpips="none", gen=model)
    ref_optim = torch.optim.Adam(model.parameters(), lr=3e-4)
    torch.save({"gen": sd, "g_optim": ref_optim.state_dict(), "args": None}, tmp_path / "004999.pt")
    assert tr.resume(str(tmp_path / "004999.pt")) == 4999
    # offline converter: HFA-GP checkpoint -> tensors for load_G_official(weights=...)
    spec = importlib.util.spec_from_file_location("convert_eg3d_pickle", os.path.join(ROOT, "tools", "convert_eg3d_pickle.py"))
    conv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(conv)
    picked = conv.select_for(model.generator, conv.hfagp_state_dict(str(tmp_path / "004999.pt")))
    assert set(picked) == set(_eg3d_keys())


def test_generator_deepcopy_and_cache_invalidation():
    """`copy.deepcopy(generator)` (what the reference's load_G_official does) keeps the precision settings and
    resolves super-resolution membership by module; the derived caches are not copied; `invalidate_caches` empties them."""
    import copy
    import torch
    from hfa_gp_amd.config import tiny64
    from hfa_gp_amd.generator import TriPlaneGenerator
    g = TriPlaneGenerator(tiny64(), seed=0)
    g.sr_conv_precision = "f16"
    g._prep[("Q", 123)] = (0, 0, None, torch.zeros(1))
    g._scalars[5] = (0, 0, 1.0)
    c = copy.deepcopy(g)
    assert c.sr_conv_precision == "f16" and c._prep == {} and c._scalars == {} and len(g._prep) == 1
    w = c.superresolution.block1.conv0.weight
    assert c._is_sr_weight(w) and not g._is_sr_weight(w) and not c._is_sr_weight(c.backbone.synthesis.b8.conv0.weight)
    g.invalidate_caches()
    assert g._prep == {} and g._scalars == {}
    c2 = c.to(torch.float32)           # _apply hook must not break module moves
    assert c2 is c

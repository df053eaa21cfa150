"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (build container only).

Run:  python tests/golden/make_golden.py          (needs /root/reference; never runs on the GPU box)

What is captured: outputs of the reference's own importable Python for seeded
inputs — data only, no reference source.  `dnnlib` / `legacy` (EG3D, not shipped) are stubbed with
empty modules and `load_G_official` is replaced by a fake generator that records what it is called
with, so that `HeadNeRF_*.__init__/get_latent/get_image/forward` run exactly as shipped.
"""
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/code"
OUT = os.path.dirname(os.path.abspath(__file__))


class FakeGenerator(torch.nn.Module):
    """Stands in for the EG3D G_ema: image = mean(ws) broadcast + label checksum (differentiable in ws)."""
    def __init__(self):
        super().__init__()
        self.dummy = torch.nn.Parameter(torch.zeros(1))
        self.calls = []

    def synthesis(self, ws, c=None, noise_mode=None):
        self.calls.append((ws.detach().clone(), c.detach().clone(), noise_mode))
        img = ws.mean(dim=(1, 2)).view(-1, 1, 1, 1).expand(-1, 3, 4, 4) + c.sum(1).view(-1, 1, 1, 1) * 1e-3
        return {"image": img}


def main():
    sys.path.insert(0, REF)
    sys.modules.setdefault("dnnlib", types.ModuleType("dnnlib"))
    sys.modules.setdefault("legacy", types.ModuleType("legacy"))
    import cam_utils
    from networks import encoder3d, headnerf
    headnerf.load_G_official = lambda args, device, *a, **k: FakeGenerator()

    g = {}
    # ---- (2) cam_utils
    hv = torch.tensor([[0.5, 0.5], [0.3, 0.5], [0.7, 0.4], [0.5, 0.7], [0.45, 0.55], [0.6, 0.6], [0.2, 0.3], [0.8, 0.65]]) * math.pi
    pts, c2w = [], []
    for h, v in hv:
        p, phi, theta = cam_utils.sample_camera_positions("cpu", n=1, r=2.7, horizontal_mean=float(h),
                                                          vertical_mean=float(v), mode=None)
        pts.append(p)
        c2w.append(cam_utils.create_cam2world_matrix(-p, p, device="cpu"))
    g["cam_hv"] = hv.numpy()
    g["cam_points"] = torch.cat(pts).numpy()
    g["cam_c2w"] = torch.cat(c2w).numpy()
    torch.manual_seed(20)
    p, phi, theta = cam_utils.sample_camera_positions("cpu", n=6, r=2.7, horizontal_stddev=0.3, vertical_stddev=0.155,
                                                      horizontal_mean=0.5 * math.pi, vertical_mean=0.5 * math.pi,
                                                      mode="gaussian")
    g["cam_gauss_points"], g["cam_gauss_phi"], g["cam_gauss_theta"] = p.numpy(), phi.numpy(), theta.numpy()
    g["cam_gauss_c2w"] = cam_utils.create_cam2world_matrix(-p, p, device="cpu").numpy()

    # ---- (4) encoder3d ops
    torch.manual_seed(1)
    x = torch.randn(1, 4, 9, 9)
    k = encoder3d.make_kernel([1, 3, 3, 1])
    g["fir_x"], g["fir_k"] = x.numpy(), k.numpy()
    for name, (up, down, pad) in {"u1d1p21": (1, 1, (2, 1)), "u1d1p11": (1, 1, (1, 1)), "u2d1p21": (2, 1, (2, 1)),
                                   "u1d2p11": (1, 2, (1, 1)), "u2d2p21": (2, 2, (2, 1)), "u1d1p0m1": (1, 1, (0, -1))}.items():
        g["fir_" + name] = encoder3d.upfirdn2d(x, k, up=up, down=down, pad=pad).numpy()
    g["fir_up2gain4"] = encoder3d.upfirdn2d(x, k * 4, up=2, down=1, pad=(2, 1)).numpy()
    b = torch.randn(1, 4, 1, 1)
    g["flrelu_b"] = b.numpy()
    g["flrelu_y"] = encoder3d.fused_leaky_relu(x, b).numpy()
    torch.manual_seed(2)
    lin = encoder3d.EqualLinear(16, 8, lr_mul=0.5, bias_init=0.3)
    xin = torch.randn(5, 16)
    g["eqlin_w"], g["eqlin_b"], g["eqlin_x"], g["eqlin_y"] = lin.weight.detach().numpy(), lin.bias.detach().numpy(), xin.numpy(), lin(xin).detach().numpy()
    conv = encoder3d.EqualConv2d(4, 6, 3, stride=1, padding=1)
    g["eqconv_w"], g["eqconv_b"] = conv.weight.detach().numpy(), conv.bias.detach().numpy()
    g["eqconv_y"] = conv(x).detach().numpy()
    torch.manual_seed(3)
    rb = encoder3d.ResBlock(8, 16)
    xr = torch.randn(2, 8, 12, 12)
    g["resblock_x"] = xr.numpy()
    g["resblock_y"] = rb(xr).detach().numpy()
    for kk, vv in rb.state_dict().items():
        g["resblock_sd/" + kk] = vv.numpy()
    # full Encoder(64) with a seeded state dict (store the seed and a few weights, not all 22 M floats)
    torch.manual_seed(4)
    enc = encoder3d.Encoder(64, 512, 50, False, False)
    xe = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(5))
    g["enc64_seed"] = np.array([4, 5])
    g["enc64_keys"] = np.array(list(enc.state_dict().keys()))
    g["enc64_shapes"] = np.array([str(tuple(v.shape)) for v in enc.state_dict().values()])
    g["enc64_y"] = enc(xe).detach().numpy()
    g["enc64_w_probe"] = enc.state_dict()["net_app.convs.1.conv1.0.weight"].flatten()[:16].numpy()

    # ---- (5) driver nets
    torch.manual_seed(6)
    w3 = headnerf.Weights_3DMM(76, 512, 50)
    xp = torch.randn(3, 76, generator=torch.Generator().manual_seed(7))
    g["w3dmm_y"] = w3(xp).detach().numpy()
    g["w3dmm_keys"] = np.array(list(w3.state_dict().keys()))
    torch.manual_seed(8)
    an = headnerf.AudioNet(64, 16)
    xa = torch.randn(8, 16, 29, generator=torch.Generator().manual_seed(9))
    g["audnet_y"] = an(xa).detach().numpy()
    g["audnet_keys"] = np.array(list(an.state_dict().keys()))
    torch.manual_seed(10)
    aa = headnerf.AudioAttNet()
    xat = torch.randn(8, 64, generator=torch.Generator().manual_seed(11))
    g["audatt_y"] = aa(xat).detach().numpy()
    g["audatt_keys"] = np.array(list(aa.state_dict().keys()))

    # ---- (1) + (3) HeadNeRF latent-basis layer and label flip
    class Args:
        out_pose = False
        person_2 = False
        params_len = 76

    for K in (50, 8):
        torch.manual_seed(12)
        m = headnerf.HeadNeRF_3DMM(Args(), 64, "cpu", 512, K)
        sd = m.state_dict()
        g[f"hn{K}_keys"] = np.array(list(sd.keys()))
        if K == 8:
            g["hn8_bases"] = sd["bases"].numpy()
            g["hn8_delta"] = sd["delta"].numpy()
        else:
            g["hn50_bases_probe"] = sd["bases"][:, :8].numpy()
            g["hn50_delta_is_mean"] = np.array(float((sd["delta"] - sd["bases"].mean(0)).abs().max()))
        alpha = torch.randn(2, K, generator=torch.Generator().manual_seed(13), requires_grad=True)
        ws = m.get_latent(alpha)
        g[f"hn{K}_alpha"] = alpha.detach().numpy()
        g[f"hn{K}_ws"] = ws.detach().numpy() if K == 8 else ws.detach().numpy()[:, :, :16]
        Q = m.get_latent(None) if False else torch.qr((m.bases + 1e-8).T)[0]
        g[f"hn{K}_QtQ_err"] = np.array(float((Q.T @ Q - torch.eye(K)).abs().max()))
        up = torch.randn(ws.shape, generator=torch.Generator().manual_seed(14))
        (ws * up).sum().backward()
        g[f"hn{K}_dalpha"] = alpha.grad.numpy()
        g[f"hn{K}_ddelta"] = m.delta.grad.numpy() if K == 8 else m.delta.grad.numpy()[:64]
        g[f"hn{K}_dbases"] = m.bases.grad.numpy() if K == 8 else m.bases.grad.numpy()[:, :64]
        if K == 8:
            g["hn8_upstream"] = up.numpy()
    # label flip through get_image / forward on the stub generator (in place, repeated calls alternate)
    torch.manual_seed(15)
    m = headnerf.HeadNeRF_3DMM(Args(), 64, "cpu", 512, 8)
    label = torch.arange(50, dtype=torch.float32).reshape(2, 25) + 1
    g["flip_label_before"] = label.clone().numpy()
    img1 = m.get_image(torch.ones(2, 14, 512), label)
    g["flip_label_after1"] = label.clone().numpy()
    g["flip_seen_by_generator1"] = m.generator.calls[-1][1].numpy()
    img2 = m.get_image(torch.ones(2, 14, 512), label)
    g["flip_label_after2"] = label.clone().numpy()
    g["flip_noise_mode"] = np.array(m.generator.calls[-1][2])
    params = torch.randn(2, 76, generator=torch.Generator().manual_seed(16))
    label3 = label.clone()
    out = m(params, label3)
    g["fwd_label_after"] = label3.numpy()
    g["fwd_ws_seen"] = m.generator.calls[-1][0].numpy()[:, :, :8]
    g["fwd_params"] = params.numpy()
    for kk, vv in m.weights_3dmm.state_dict().items():
        pass  # weights are seed-defined (seed 15); not stored

    # ---- (6) layout_grid uint8 quantisation: the reference's own function
    # (run_recon_video_rgb.py:28-42); its unrelated top-level imports (torchvision, imageio, dataset)
    # are absent here and are replaced by mock modules just to get the module object.
    from unittest import mock
    for name in ("torchvision", "torchvision.transforms", "imageio", "dataset"):
        sys.modules.setdefault(name, mock.MagicMock())
    import run_recon_video_rgb as rr
    ramp = torch.linspace(-1.2, 1.2, 2 * 3 * 4 * 5).reshape(2, 3, 4, 5)
    g["grid_in"] = ramp.numpy()
    g["grid_out"] = rr.layout_grid(ramp, grid_w=2, grid_h=1)

    np.savez_compressed(os.path.join(OUT, "reference_vectors.npz"), **g)
    print("wrote", os.path.join(OUT, "reference_vectors.npz"), len(g), "arrays")


if __name__ == "__main__":
    with torch.no_grad() if False else torch.enable_grad():
        main()

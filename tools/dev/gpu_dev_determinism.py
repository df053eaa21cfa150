"""Developer check (GPU box): bit-for-bit repeatability of the forward path — the same synthesis call N times, every
intermediate the generator exposes compared with the first run (planes, feature image, raw image, final image).  A
sporadic hardware / code-generation hazard (see csrc/torgb_skip.hip) shows up here as isolated differing elements."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hfa_gp_amd.config import ffhq512_128
from hfa_gp_amd.generator import TriPlaneGenerator
from hfa_gp_amd.synthetic import make_inputs, perturb_state


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    cfg = ffhq512_128()
    gen = perturb_state(TriPlaneGenerator(cfg, seed=0)).requires_grad_(False).cuda()
    for B in (1, 4):
        ws, c, us, ui = (t.cuda() for t in make_inputs(cfg, B, seed=5))
        for prec, srp, store in (("f16x3", None, "f32"), ("bf16x3", None, "f32"), ("f16x3", "f16", "f32"), ("f16x3", "f16", "f16"),
                                 ("fp32", None, "f32")):
            gen.conv_precision, gen.sr_conv_precision, gen.sr_storage = prec, srp, store
            ref, bad = None, {}
            for t in range(n):
                with torch.no_grad():
                    out = gen.synthesis(ws, c, u_strat=us, u_imp=ui, return_planes=True)
                cur = {k: v.clone() for k, v in out.items() if torch.is_tensor(v)}
                if ref is None:
                    ref = cur
                    continue
                for k, v in cur.items():
                    d = int((v != ref[k]).sum().item())
                    if d:
                        bad[k] = bad.get(k, 0) + d
            print(f"B={B} conv {prec} sr {srp} storage {store}: {n} runs, differing elements {bad if bad else 'none'} (keys {sorted(ref)})", flush=True)

    # ---- backward: the ray marcher's scatter uses fp32 atomics (order-dependent rounding by design), everything else is
    # meant to be bit-repeatable.  With the scatter replaced by a FIXED d_planes (and its input g_feat recorded) the whole
    # backward pass — super-resolution data / style / weight gradients, backbone likewise — must repeat bit for bit.
    from hfa_gp_amd import ops
    real_bwd = ops.raymarch_bwd
    seen = {}

    def fake_bwd(g_feat, planes, *a, decoder_grads=False, **kw):
        seen["g_feat"] = g_feat.clone()
        gg = torch.Generator(device="cuda").manual_seed(7)
        d_planes = torch.randn(planes.shape, generator=gg, device="cuda") * 1e-3
        if decoder_grads:
            return d_planes, tuple(torch.zeros_like(t) for t in (kw["dec_w0"], kw["dec_b0"], kw["dec_w1"], kw["dec_b1"]))
        return d_planes

    ops.raymarch_bwd = fake_bwd
    gen.conv_precision, gen.sr_conv_precision, gen.sr_storage = "f16x3", None, "f32"
    try:
        for tuned in (False, True):
            gen.requires_grad_(tuned)
            B = 2
            ws, c, us, ui = (t.cuda() for t in make_inputs(cfg, B, seed=6))
            gimg = torch.randn(B, 3, 512, 512, device="cuda")
            ref, bad = None, {}
            for t in range(max(4, n // 3)):
                wsg = ws.clone().requires_grad_(True)
                for p_ in gen.parameters():
                    p_.grad = None
                img = gen.synthesis(wsg, c, u_strat=us, u_imp=ui)["image"]
                (img * gimg).sum().backward()
                cur = {"d_ws": wsg.grad.clone(), "g_feat": seen["g_feat"]}
                if tuned:
                    for k, p_ in gen.named_parameters():
                        if p_.grad is not None:
                            cur[k] = p_.grad.clone()
                if ref is None:
                    ref = cur
                    continue
                for k, v in cur.items():
                    d = int((v != ref[k]).sum().item())
                    if d:
                        bad[k] = bad.get(k, 0) + d
            print(f"backward (scatter replaced), generator {'tuned' if tuned else 'frozen'}: {len(ref)} tensors, "
                  f"differing elements {bad if bad else 'none'}", flush=True)
    finally:
        ops.raymarch_bwd = real_bwd
        gen.requires_grad_(False)


if __name__ == "__main__":
    main()

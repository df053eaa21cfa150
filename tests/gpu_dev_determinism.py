"""Developer check (GPU box): bit-for-bit repeatability of the forward path — the same synthesis call N times, every
intermediate the generator exposes compared with the first run (planes, feature image, raw image, final image).  A
sporadic hardware / code-generation hazard (see csrc/torgb_skip.hip) shows up here as isolated differing elements."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hfa_gp_amd.config import ffhq512_128
from hfa_gp_amd.generator import TriPlaneGenerator
from tests.util import make_inputs, perturb_state


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


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Offline converter: EG3D network pickle -> safetensors with the key names `hfa_gp_amd.generator` loads.

HFA-GP loads `./code/pretrained_models/eg3d/ffhqrebalanced512-128.pkl` with NVlabs/eg3d's `legacy.load_network_pkl`
(/root/reference/code/networks/headnerf.py:31-38); neither the pickle nor EG3D ships with the reference, so
this script is meant to be run ONCE by a user who has both (it needs `dnnlib` and `legacy` from the EG3D
checkout on PYTHONPATH because the pickle embeds EG3D's classes):

    PYTHONPATH=/path/to/eg3d/eg3d python tools/convert_eg3d_pickle.py ffhqrebalanced512-128.pkl ffhq512-128.safetensors

`TriPlaneGenerator` in this repo registers its parameters and buffers under EG3D's own names
(`backbone.synthesis.b64.conv1.affine.weight`, `superresolution.block1.torgb.bias`, `decoder.net.2.weight`,
`backbone.mapping.fc0.weight`, ...), so the conversion is: take `G_ema.state_dict()`, keep the keys the
generator has, check shapes, write fp32 safetensors.  It can also unwrap an HFA-GP training checkpoint
(`ckpt["gen"]`, keys prefixed `generator.`; /root/reference/code/trainer_rgb.py:130-151) with `--hfagp-ckpt`.

This file has no counterpart on the GPU box and is not imported by the package.
"""
from __future__ import annotations

import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def eg3d_state_dict(path: str) -> dict:
    import dnnlib   # noqa: F401  (EG3D; required to unpickle)
    import legacy   # EG3D
    with dnnlib.util.open_url(path) as f:
        g = legacy.load_network_pkl(f)["G_ema"]
    return {k: v.detach().float().cpu() for k, v in g.state_dict().items()}


def hfagp_state_dict(path: str) -> dict:
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    gen = ckpt["gen"] if "gen" in ckpt else ckpt
    return {k[len("generator."):]: v.detach().float().cpu() for k, v in gen.items() if k.startswith("generator.")}


def select_for(gen, src: dict, strict: bool = True) -> dict:
    """Keys of ``gen`` taken from ``src``; raises on a missing key or a shape mismatch."""
    want = gen.state_dict()
    out, missing, bad = {}, [], []
    for k, ref in want.items():
        if k not in src:
            missing.append(k)
            continue
        v = src[k]
        if tuple(v.shape) != tuple(ref.shape):
            bad.append((k, tuple(v.shape), tuple(ref.shape)))
            continue
        out[k] = v.contiguous()
    if bad:
        raise SystemExit(f"shape mismatch (config does not describe this checkpoint): {bad[:5]}")
    if missing and strict:
        raise SystemExit(f"{len(missing)} keys missing from the checkpoint, e.g. {missing[:5]}")
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--preset", default="ffhq512_128")
    ap.add_argument("--hfagp-ckpt", action="store_true", help="src is an HFA-GP checkpoint (ckpt['gen'])")
    args = ap.parse_args()
    from safetensors.torch import save_file
    from hfa_gp_amd.config import PRESETS
    from hfa_gp_amd.generator import TriPlaneGenerator
    gen = TriPlaneGenerator(PRESETS[args.preset]())
    src = hfagp_state_dict(args.src) if args.hfagp_ckpt else eg3d_state_dict(args.src)
    tensors = select_for(gen, src)
    extra = sorted(set(src) - set(tensors))
    save_file(tensors, args.dst)
    print(f"wrote {len(tensors)} tensors to {args.dst}; ignored {len(extra)} source keys"
          + (f" (e.g. {extra[:4]})" if extra else ""))


if __name__ == "__main__":
    main()

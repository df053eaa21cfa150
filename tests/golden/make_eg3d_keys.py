#!/usr/bin/env python
"""Writes tests/golden/eg3d_ffhq512_128_keys.json: names and shapes (no values) of `G_ema.state_dict()` of NVlabs/eg3d's
`ffhqrebalanced512-128.pkl`, the checkpoint HFA-GP loads (/root/reference/code/networks/headnerf.py:31-38), and of the
`generator.*` part of an HFA-GP training checkpoint `ckpt["gen"]` (/root/reference/code/trainer_rgb.py:143-151).

PROVENANCE: neither EG3D nor the pickle is available offline, so this list is ENUMERATED FROM THE
MODULE STRUCTURE of EG3D's `TriPlaneGenerator` as recalled ([EG3D-recall]) — NOT captured from
the real pickle.  It is written independently of `hfa_gp_amd.generator` (plain loops below, no import of the package) so
that the test comparing the two is not a tautology; a user with the real pickle can regenerate it with
    python -c "import json,legacy,dnnlib; g=legacy.load_network_pkl(dnnlib.util.open_url(P))['G_ema']; \
               json.dump({k:list(v.shape) for k,v in g.state_dict().items()}, open(OUT,'w'), indent=0)"
and the tests will then check the package against the real key set.

Structure enumerated (training/triplane.py, networks_stylegan2.py, superresolution.py, EG3D):
  backbone  = StyleGAN2Backbone(z 512, c 25, w 512, img_resolution 256, img_channels 96, mapping num_layers 2)
      .synthesis.b{4..256}: SynthesisBlock  [const | conv0(up 2)] conv1 torgb + buffer resample_filter[4,4]
          SynthesisLayer: weight[Co,Ci,3,3] noise_strength[] bias[Co] buffers resample_filter[4,4] noise_const[r,r] affine.{weight[Ci,512],bias[Ci]}
          ToRGBLayer:     weight[96,Co,1,1] bias[96] affine.{weight[Co,512],bias[Co]}
      .mapping: embed.{weight[512,25],bias[512]} fc0.{weight[512,1024],bias} fc1.{weight[512,512],bias} buffer w_avg[512]
  superresolution = SuperresolutionHybrid8XDC: block0 = SynthesisBlock(32->256 @256, img 3), block1 = (256->128 @512, img 3)
  decoder   = OSGDecoder: net.0 = FullyConnectedLayer(32,64), net.2 = FullyConnectedLayer(64,33)
  renderer / ray_sampler: no parameters, no buffers
"""
import json
import os

W = 512


def layer(prefix, ci, co, res, out):
    out[prefix + ".weight"] = [co, ci, 3, 3]
    out[prefix + ".noise_strength"] = []
    out[prefix + ".bias"] = [co]
    out[prefix + ".resample_filter"] = [4, 4]
    out[prefix + ".noise_const"] = [res, res]
    out[prefix + ".affine.weight"] = [ci, W]
    out[prefix + ".affine.bias"] = [ci]


def torgb(prefix, ci, img, out):
    out[prefix + ".weight"] = [img, ci, 1, 1]
    out[prefix + ".bias"] = [img]
    out[prefix + ".affine.weight"] = [ci, W]
    out[prefix + ".affine.bias"] = [ci]


def block(prefix, ci, co, res, img, out):
    if ci == 0:
        out[prefix + ".const"] = [co, res, res]
    else:
        layer(prefix + ".conv0", ci, co, res, out)
    out[prefix + ".resample_filter"] = [4, 4]
    layer(prefix + ".conv1", co, co, res, out)
    torgb(prefix + ".torgb", co, img, out)


def main():
    out = {}
    nf = lambda res: min(32768 // res, 512)
    for res in (4, 8, 16, 32, 64, 128, 256):
        block(f"backbone.synthesis.b{res}", nf(res // 2) if res > 4 else 0, nf(res), res, 96, out)
    out["backbone.mapping.embed.weight"] = [W, 25]
    out["backbone.mapping.embed.bias"] = [W]
    out["backbone.mapping.fc0.weight"] = [W, 2 * W]
    out["backbone.mapping.fc0.bias"] = [W]
    out["backbone.mapping.fc1.weight"] = [W, W]
    out["backbone.mapping.fc1.bias"] = [W]
    out["backbone.mapping.w_avg"] = [W]
    block("superresolution.block0", 32, 256, 256, 3, out)
    block("superresolution.block1", 256, 128, 512, 3, out)
    out["decoder.net.0.weight"] = [64, 32]
    out["decoder.net.0.bias"] = [64]
    out["decoder.net.2.weight"] = [33, 64]
    out["decoder.net.2.bias"] = [33]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "eg3d_ffhq512_128_keys.json")
    json.dump({"provenance": "enumerated from EG3D's module structure as recalled; NOT captured from the real pickle "
                             "(see make_eg3d_keys.py)", "keys": out}, open(path, "w"), indent=0, sort_keys=True)
    print(len(out), "keys,", sum(max(1, eval("*".join(map(str, s)) or "1")) for s in out.values()), "elements")


if __name__ == "__main__":
    main()

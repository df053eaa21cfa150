"""Generator configuration for the EG3D tri-plane hot path.

Every constant that SURVEY.md §10 (uncertainty register U1-U12) lists as
"recalled from NVlabs/eg3d" is a field here, so that a mismatch against a real
EG3D checkpoint is a config fix, not a rewrite.  The reference only reaches the
generator through ``generator.synthesis(ws, c=label, noise_mode='const')``
(/root/reference/code/networks/headnerf.py:112,118,133) and feeds it
``ws[B,14,512]`` (headnerf.py:55) and ``label[B,25]``.
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import Dict, List, Optional, Tuple


@dataclass
class GeneratorConfig:
    # ---- latent / label -------------------------------------------------
    w_dim: int = 512
    c_dim: int = 25
    # ---- StyleGAN2 backbone (tri-plane synthesis) ------------------------
    plane_resolution: int = 256          # backbone img_resolution
    plane_channels: int = 32             # per plane; backbone img_channels = 3*32
    channel_base: int = 32768
    channel_max: int = 512
    backbone_conv_clamp: Optional[float] = None   # U4: num_fp16_res 0 -> None
    backbone_noise_mode: str = "const"            # headnerf.py:112 passes noise_mode='const'
    # ---- neural renderer ---------------------------------------------------
    neural_rendering_resolution: int = 128
    ray_start: float = 2.25
    ray_end: float = 3.3
    depth_resolution: int = 48
    depth_resolution_importance: int = 48
    box_warp: float = 1.0
    white_back: bool = False
    plane_axes: str = "eg3d_original"    # U1: (x,y),(x,z),(z,x); "eg3d_fixed": (x,y),(x,z),(z,y)
    decoder_hidden: int = 64
    decoder_lr_mul: float = 1.0
    # ---- super-resolution (SuperresolutionHybrid8XDC topology) -------------
    img_resolution: int = 512
    sr_channels: Tuple[int, int] = (256, 128)     # block0 out, block1 out
    sr_conv_clamp: Optional[float] = 256.0        # U4: sr_num_fp16_res>0 -> 256 (applied in fp32 too)
    sr_noise_mode: str = "none"
    img_channels: int = 3
    # ---- shared op constants -------------------------------------------------
    resample_filter: Tuple[int, ...] = (1, 3, 3, 1)
    lrelu_alpha: float = 0.2
    demod_eps: float = 1e-8
    # arithmetic of the conv GEMMs (include/hfagp.h HFAGP_PREC_*): "fp32" = exact v_mfma_f32_32x32x2_f32;
    # "bf16x3" = operands split into hi+lo bf16, 3 bf16 MFMAs per product, fp32 accumulation (relative error of a
    # layer ~5e-6, i.e. ~200x below the fp16 the reference's CUDA path runs the super-resolution blocks in, U4);
    # "bf16x6" = 3 parts / 6 MFMAs (fp32-class).  Layers the split kernel cannot take run on the exact kernel.
    # "f16x3" (default) = operands split into hi+lo fp16 (11 + 11 mantissa bits), 3 fp16 MFMAs per product: relative
    # error of a layer ~1e-6 = the exact kernel's own summation noise, ~5 % slower than "bf16x3".
    # "f16x2" (opt-in) = the f16x3 weight image against activations rounded to ONE fp16 part, 2 MFMAs per product: the TF32
    # class the reference's cuDNN convs run in (relative error of a layer ~1e-4), +18 % frames/s.
    # "f16" = operands rounded to fp16, ONE fp16 MFMA per product, fp32 accumulation: the arithmetic of EG3D's own
    # fp16 blocks (relative error of a layer ~3e-4); never the default.
    conv_precision: str = "f16x3"
    # precision of the two super-resolution blocks when it differs from conv_precision: "f16" reproduces the
    # reference's CUDA defaults (fp32 backbone, fp16 super-resolution: sr_num_fp16_res = 4)
    sr_conv_precision: Optional[str] = None
    # storage of the activations BETWEEN the super-resolution layers: "f32" (default) or "f16" — with sr_conv_precision
    # "f16", forward-only calls keep them in fp16 as EG3D's fp16 blocks do (half the HBM traffic of these HBM-heavy
    # layers; values are rounded to fp16 exactly where the reference's `x.to(torch.float16)` tensors are)
    sr_storage: str = "f32"
    # arithmetic of the decoder MLP inside the ray marcher: "f16x3" = split fp16 operands on the 16-bit matrix pipe
    # (3 MFMAs per product, ~2^-22: fp32-class; operands scaled into fp16's range by exact powers of two from a bound on
    # |planes| that the plane-writing kernel publishes), "fp32" = the exact fp32 matrix instructions (5x the pipe time)
    decoder_precision: str = "f16x3"
    # ---- mapping network (never called by HFA-GP) ---------------
    z_dim: int = 512
    mapping_layers: int = 2
    mapping_lr_mul: float = 0.01
    name: str = "ffhq512_128"

    # -- derived ---------------------------------------------------------------
    @property
    def block_resolutions(self) -> List[int]:
        r, out = 4, []
        while r <= self.plane_resolution:
            out.append(r)
            r *= 2
        return out

    def channels(self, res: int) -> int:
        return min(self.channel_base // res, self.channel_max)

    @property
    def backbone_img_channels(self) -> int:
        return 3 * self.plane_channels

    @property
    def num_ws(self) -> int:
        # b4 has one conv, every other block two; the last block's toRGB adds one.
        n = 0
        for res in self.block_resolutions:
            n += 1 if res == 4 else 2
        return n + 1

    @property
    def sr_resolutions(self) -> Tuple[int, int]:
        return (self.neural_rendering_resolution * 2, self.neural_rendering_resolution * 4)

    @property
    def samples_per_ray(self) -> int:
        return self.depth_resolution + self.depth_resolution_importance

    def to_dict(self) -> Dict:
        return asdict(self)

    def validate(self) -> None:
        assert self.img_resolution == self.neural_rendering_resolution * 4, \
            "SR topology is two up-2 blocks (SuperresolutionHybrid8XDC)"
        assert self.depth_resolution % 16 == 0 and self.depth_resolution <= 64
        assert self.depth_resolution_importance % 16 == 0 and self.depth_resolution_importance <= 64
        assert self.plane_channels == 32, "decoder / ray-march kernel are written for 32 features"
        assert self.plane_axes in ("eg3d_original", "eg3d_fixed")
        assert self.decoder_precision in ("fp32", "f16x3")
        assert self.sr_storage in ("f32", "f16")


def ffhq512_128() -> GeneratorConfig:
    """BASELINE configs 2-5: the FFHQ 512-128 EG3D generator HFA-GP loads
    (headnerf.py:31, 'ffhqrebalanced512-128.pkl')."""
    return GeneratorConfig()


def tiny64() -> GeneratorConfig:
    """BASELINE config 1 (plumbing): same topology at 1/8 scale, 16+16 samples."""
    return GeneratorConfig(
        plane_resolution=64, channel_base=2048, channel_max=64,
        neural_rendering_resolution=16, depth_resolution=16,
        depth_resolution_importance=16, img_resolution=64,
        sr_channels=(64, 32), name="tiny64")


def small128() -> GeneratorConfig:
    """Mid-size parity case: 128^2 image, 32^2 rays, 32+32 samples."""
    return GeneratorConfig(
        plane_resolution=128, channel_base=8192, channel_max=128,
        neural_rendering_resolution=32, depth_resolution=32,
        depth_resolution_importance=32, img_resolution=128,
        sr_channels=(64, 32), name="small128")


def tiny14() -> GeneratorConfig:
    """Smallest preset that keeps HFA-GP's 14-row W+ latent (plane resolution 256 -> 7 blocks):
    used to test the HeadNeRF_* boundary, which hard-codes 14 x 512 (headnerf.py:55)."""
    return GeneratorConfig(
        plane_resolution=256, channel_base=2048, channel_max=32,
        neural_rendering_resolution=16, depth_resolution=16,
        depth_resolution_importance=16, img_resolution=64,
        sr_channels=(32, 16), name="tiny14")


PRESETS = {"ffhq512_128": ffhq512_128, "tiny64": tiny64, "small128": small128, "tiny14": tiny14}

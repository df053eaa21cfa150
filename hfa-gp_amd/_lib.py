"""ctypes binding of ``libhfagp_hip.so`` (C ABI declared in ``include/hfagp.h``).

The product path has NO fallback: if the shared library is missing or its ABI
version differs, ``lib()`` raises ``RuntimeError`` (build it with
``python -c "import __graft_entry__ as g; g.build()"`` or
``bash hfa-gp_amd/csrc/build.sh``).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# (HFAGP_LIB_PATH: developer override, used by the ablation builds of tools/dev/ — the product loads the in-tree library)
LIB_PATH = os.environ.get("HFAGP_LIB_PATH") or os.path.join(_HERE, "libhfagp_hip.so")
ABI_VERSION = 12

c_float_p = C.c_void_p  # device pointers travel as integers


class RaymarchArgs(C.Structure):
    _fields_ = [
        ("planes", C.c_void_p), ("cam2world", C.c_void_p), ("intrinsics", C.c_void_p),
        ("u_strat", C.c_void_p), ("u_imp", C.c_void_p),
        ("dec_w0", C.c_void_p), ("dec_b0", C.c_void_p), ("dec_w1", C.c_void_p), ("dec_b1", C.c_void_p),
        ("feat", C.c_void_p), ("depth", C.c_void_p), ("wsum", C.c_void_p), ("tminmax", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("res", C.c_int32),
        ("Sc", C.c_int32), ("Sf", C.c_int32), ("plane_axes", C.c_int32), ("white_back", C.c_int32),
        ("ray_start", C.c_double), ("ray_end", C.c_double),
        ("box_warp", C.c_float), ("decoder_lr_mul", C.c_float),
        ("planes_absmax", C.c_void_p), ("state", C.c_void_p),
    ]


class StyleArgs(C.Structure):
    _fields_ = [
        ("w", C.c_void_p), ("affine_w", C.c_void_p), ("affine_b", C.c_void_p), ("wsq", C.c_void_p),
        ("styles", C.c_void_p), ("dcoef", C.c_void_p),
        ("B", C.c_int32), ("w_dim", C.c_int32), ("w_stride", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
        ("style_gain", C.c_float), ("eps", C.c_float),
    ]


class ModconvArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("wt", C.c_void_p), ("styles", C.c_void_p), ("dcoef", C.c_void_p),
        ("noise", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p), ("workspace", C.c_void_p),
        ("x_batch_stride", C.c_int64),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
        ("mode", C.c_int32), ("act", C.c_int32), ("ksplit", C.c_int32),
        ("noise_strength", C.c_float), ("alpha", C.c_float), ("gain", C.c_float), ("clamp", C.c_float),
        ("precision", C.c_int32),
        ("x_absmax", C.c_void_p), ("y_absmax", C.c_void_p),
        ("rgb_w", C.c_void_p), ("rgb_part", C.c_void_p),
        ("x_f16", C.c_int32), ("y_f16", C.c_int32),
    ]


ABSMAX_SLOTS = 64       # HFAGP_ABSMAX_SLOTS
ABSMAX_FLOATS = 64 * 32  # HFAGP_ABSMAX_FLOATS: one 128-byte line per slot


class UpfirEpilogueArgs(C.Structure):
    _fields_ = [
        ("yt", C.c_void_p), ("dcoef", C.c_void_p), ("noise", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("act", C.c_int32),
        ("noise_strength", C.c_float), ("alpha", C.c_float), ("gain", C.c_float), ("clamp", C.c_float),
        ("y_absmax", C.c_void_p), ("io_f16", C.c_int32),
    ]


class TorgbSkipArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("wt", C.c_void_p), ("styles", C.c_void_p), ("bias", C.c_void_p),
        ("img_in", C.c_void_p), ("img_out", C.c_void_p), ("x_absmax", C.c_void_p), ("out_absmax", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
        ("precision", C.c_int32), ("plane_major", C.c_int32),
    ]


class SkipArgs(C.Structure):
    _fields_ = [
        ("img_in", C.c_void_p), ("y", C.c_void_p), ("img_out", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("plane_major", C.c_int32),
        ("out_absmax", C.c_void_p),
    ]


class TorgbArgs(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("weight", C.c_void_p), ("styles", C.c_void_p), ("bias", C.c_void_p),
        ("rgb_in", C.c_void_p), ("rgb_out", C.c_void_p), ("y_pre", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
        ("clamp", C.c_float),
    ]


class TorgbFinishArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("part", "bias", "rgb_in", "rgb_out", "y_pre")] + \
        [(n, C.c_int32) for n in ("nparts", "B", "H", "W", "Cout")] + [("clamp", C.c_float)]


class PointwiseBwdArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "dxs_conv", "s_conv", "dxs_rgb", "s_rgb", "g_rgb_small", "w_rgb_small", "s_small", "g_direct", "x",
        "dcoef_p", "bias_p", "noise_p", "g_out", "partial", "sums")] + \
        [(n, C.c_int32) for n in ("B", "H", "W", "C", "Co", "nchunks", "has_producer", "act_p", "param_grads")] + \
        [(n, C.c_float) for n in ("noise_strength_p", "alpha", "gain", "clamp")] + \
        [(n, C.c_void_p) for n in ("y_rgb_small", "g_nchw3_a", "g_nchw3_b")] + [("clamp_rgb_small", C.c_float)] + \
        [("noise_strength_dev", C.c_void_p)]


class StyleBwdArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ds", "dd", "styles", "dcoef", "wsq", "affine_w", "dstot", "dw")] + \
        [(n, C.c_int32) for n in ("B", "Cin", "Cout", "w_dim", "dw_stride", "accumulate")] + \
        [("style_gain", C.c_float)]


class WgradArgs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("x", "styles", "g", "weight", "dd", "dcoef", "dweight", "workspace")] + \
        [(n, C.c_int32) for n in ("B", "H", "W", "Cin", "Cout", "mode", "ksplit", "precision", "accumulate", "dd_stride")]


class AffineGradItem(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dstot", "w", "dA", "db")] + [(n, C.c_int32) for n in ("B", "Cin", "w_dim", "w_stride")]


class ReducePartialsItem(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("sums", C.c_void_p)] + [(n, C.c_int32) for n in ("B", "nchunks", "n")]


class BiasNoiseGradItem(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("sums", "dbias", "dnoise")] + [(n, C.c_int32) for n in ("B", "C")]


class WeightPrepItem(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("weight", "image", "image_t", "wsq")] + \
        [(n, C.c_int32) for n in ("Cout", "Cin", "taps", "precision", "precision_t")]


class StyleBwdItem(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ds", "dd", "styles", "dcoef", "wsq", "affine_w", "dstot", "dw")] + \
        [(n, C.c_int32) for n in ("B", "Cin", "Cout", "w_dim", "dw_stride", "ds_stride", "dd_stride")] + \
        [("style_gain", C.c_float)]


class RaymarchBwdArgs(C.Structure):
    _fields_ = [("fwd", RaymarchArgs), ("g_feat", C.c_void_p), ("d_planes", C.c_void_p), ("rec", C.c_void_p),
                ("d_dec_w0", C.c_void_p), ("d_dec_b0", C.c_void_p), ("d_dec_w1", C.c_void_p), ("d_dec_b1", C.c_void_p),
                ("df_scratch", C.c_void_p), ("rows_scratch", C.c_void_p), ("rows_scratch_bytes", C.c_uint64)]


# every symbol include/hfagp.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "hfagp_abi_version": (C.c_int, []),
    "hfagp_last_error": (C.c_char_p, []),
    "hfagp_raymarch_fwd": (C.c_int, [C.POINTER(RaymarchArgs), C.c_void_p]),
    "hfagp_depth_clamp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "hfagp_style_fwd": (C.c_int, [C.POINTER(StyleArgs), C.c_void_p]),
    "hfagp_fc_fwd": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 3 + [C.c_float, C.c_int32, C.c_float, C.c_float, C.c_void_p]),
    "hfagp_weight_prep": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hfagp_qr_gram_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "hfagp_qr_refine_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "hfagp_tall_gram_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "hfagp_tall_gram": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                  C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "hfagp_style_batch_fwd": (C.c_int, [C.POINTER(StyleArgs), C.c_int32, C.c_void_p]),
    "hfagp_weight_prep_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hfagp_weight_prep_prec": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hfagp_modconv_workspace_bytes": (C.c_size_t, [C.POINTER(ModconvArgs)]),
    "hfagp_modconv_fwd": (C.c_int, [C.POINTER(ModconvArgs), C.c_void_p]),
    "hfagp_upfir_epilogue_fwd": (C.c_int, [C.POINTER(UpfirEpilogueArgs), C.c_void_p]),
    "hfagp_upconv_fir_scratch_bytes": (C.c_size_t, [C.POINTER(ModconvArgs)]),
    "hfagp_upconv_fir_fwd": (C.c_int, [C.POINTER(ModconvArgs), C.c_void_p, C.c_void_p]),
    "hfagp_skip_upsample_add": (C.c_int, [C.POINTER(SkipArgs), C.c_void_p]),
    "hfagp_torgb_skip_fwd": (C.c_int, [C.POINTER(TorgbSkipArgs), C.c_void_p]),
    "hfagp_blur_down_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hfagp_blur_down_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hfagp_torgb_fwd": (C.c_int, [C.POINTER(TorgbArgs), C.c_void_p]),
    "hfagp_torgb_finish_fwd": (C.c_int, [C.POINTER(TorgbFinishArgs), C.c_void_p]),
    "hfagp_modconv_rgb_parts": (C.c_int32, [C.POINTER(ModconvArgs)]),
    "hfagp_pointwise_bwd": (C.c_int, [C.POINTER(PointwiseBwdArgs), C.c_void_p]),
    "hfagp_upfir_bwd": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p]),
    "hfagp_upsample2d_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hfagp_planes_to_nhwc": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p]),
    "hfagp_style_bwd": (C.c_int, [C.POINTER(StyleBwdArgs), C.c_void_p]),
    "hfagp_style_batch_bwd": (C.c_int, [C.POINTER(StyleBwdItem), C.c_int32, C.c_void_p]),
    "hfagp_raymarch_bwd": (C.c_int, [C.POINTER(RaymarchBwdArgs), C.c_void_p]),
    "hfagp_raymarch_bwd_rows_bytes": (C.c_size_t, [C.POINTER(RaymarchArgs)]),
    "hfagp_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]),
    "hfagp_adam_chunk": (C.c_int32, []),
    "hfagp_weight_prep_batch": (C.c_int, [C.POINTER(WeightPrepItem), C.c_int32, C.c_void_p]),
    "hfagp_wgrad_ksplit": (C.c_int32, [C.POINTER(WgradArgs)]),
    "hfagp_wgrad_workspace_bytes": (C.c_size_t, [C.POINTER(WgradArgs)]),
    "hfagp_conv_wgrad": (C.c_int, [C.POINTER(WgradArgs), C.c_void_p]),
    "hfagp_affine_grad": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p]),
    "hfagp_affine_grad_batch": (C.c_int, [C.POINTER(AffineGradItem), C.c_int32, C.c_void_p]),
    "hfagp_bias_noise_grads": (C.c_int, [C.POINTER(BiasNoiseGradItem), C.c_int32, C.c_void_p]),
    "hfagp_reduce_partials_batch": (C.c_int, [C.POINTER(ReducePartialsItem), C.c_int32, C.c_void_p]),
    "hfagp_channel_sum": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hfagp_pool_mse_workspace_bytes": (C.c_size_t, []),
    "hfagp_pool_mse_fwd": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 4 + [C.c_void_p]),
    "hfagp_pool_mse_bwd": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p]),
    "hfagp_upfirdn2d_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 12 + [C.c_float, C.c_void_p]),
    "hfagp_upfirdn2d_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int32] * 12 + [C.c_float, C.c_void_p]),
    "hfagp_bias_act_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_float,
                                     C.c_float, C.c_void_p]),
    "hfagp_bias_act_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64,
                                     C.c_int32, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "hfagp_allreduce_f32": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int32, C.c_void_p]),
    "hfagp_nchw_to_nhwc": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p]),
    "hfagp_nhwc_to_nchw": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p]),
}

_lock = threading.Lock()
_lib = None


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raise loudly if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"hfa_gp_amd: HIP library not found at {LIB_PATH}. There is no CPU fallback; build it with "
                f"`bash {os.path.join(_HERE, 'csrc', 'build.sh')}` (hipcc --offload-arch=gfx950).")
        try:
            # PyTorch-ROCm ships its own libamdhip64: it must be in the process BEFORE this library is loaded, or the
            # dlopen pulls a second HIP runtime from /opt/rocm and launches on torch's streams fail with
            # "no ROCm-capable device is detected"
            import torch  # noqa: F401
            handle = C.CDLL(LIB_PATH)
        except OSError as e:  # missing libamdhip64 etc.
            raise RuntimeError(f"hfa_gp_amd: cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise RuntimeError(f"hfa_gp_amd: {LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        ver = handle.hfagp_abi_version()
        if ver != ABI_VERSION:
            raise RuntimeError(f"hfa_gp_amd: ABI version mismatch: library {ver}, binding {ABI_VERSION}")
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().hfagp_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else 'unknown error'}")

// Split-operand helpers of the 16-bit matrix-pipe kernels (modconv_bf16.hip, torgb_skip.hip): operand kinds, the split of an
// fp32 value into 16-bit parts, the MFMA wrapper and the fp16 range guard.
#pragma once
#include "common.h"

namespace hfagp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int CKB = 16;          // channels per K chunk
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ unsigned pack_f16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}

// operand kinds KD of the kernels below: 1 = one fp16 part (HFAGP_PREC_F16), 2 / 3 = two / three bf16 parts
// (BF16X3 / BF16X6), 4 = two fp16 parts (F16X3: 11 + 11 mantissa bits, the three products above 2^-22),
// 5 = F16X2: the WEIGHTS as two fp16 parts (the F16X3 image), the ACTIVATIONS as one — two MFMAs per product, an 11-bit
// activation against a 22-bit weight: the class of TF32 (11 + 11 bits), which is what the reference's cuDNN convolutions run
// in on Ampere-class GPUs (its scripts leave torch.backends.cudnn.allow_tf32 at its default).
constexpr int kind_parts(int kd) { return kd == 4 || kd == 5 ? 2 : kd; }       // parts of the weight image (B operand)
constexpr int kind_parts_a(int kd) { return kd == 5 ? 1 : kind_parts(kd); }     // parts of the activations (A operand)
constexpr int kind_split(int kd) { return kd == 5 ? 1 : kd; }                   // split4<> kind of the activations
constexpr bool kind_f16(int kd) { return kd == 1 || kd == 4 || kd == 5; }
// part products in issue order (A part, B part): 1 | a0b0 a1b0 a0b1 | + a1b1 a2b0 a0b2 | F16X2: a0b0 a0b1
constexpr int kind_nprod(int kd) { return kd == 1 ? 1 : kd == 5 ? 2 : kind_parts(kd) == 2 ? 3 : 6; }
constexpr int kind_pa(int kd, int i) { return kd == 5 ? 0 : (i == 1 || i == 3) ? 1 : i == 4 ? 2 : 0; }
constexpr int kind_pb(int kd, int i) { return kd == 5 ? i : (i == 2 || i == 3) ? 1 : i == 5 ? 2 : 0; }

__device__ __forceinline__ float f16_lo_back(unsigned u) {       // fp16 in bits 0-15 -> float
    return (float)__builtin_bit_cast(f16x2, u)[0];
}
__device__ __forceinline__ float f16_hi_back(unsigned u) {       // fp16 in bits 16-31 -> float
    return (float)__builtin_bit_cast(f16x2, u)[1];
}

// one MFMA of the path: fp16 or bf16 operands
template <bool F16>
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// v (4 floats) -> parts x 4 elements (two dwords per part), each part the round-to-nearest 16-bit value of the
// residual left by the parts before it
template <int KD>
__device__ __forceinline__ void split4(float4 v, uint2 (&out)[kind_parts_a(KD)]) {
    constexpr int NP = kind_parts_a(KD);
    if constexpr (kind_f16(KD)) {
        // saturating: an activation beyond fp16's range (never seen; the reference clamps its fp16 layers at 256) must
        // not turn into inf - inf = NaN in the residual
        constexpr float FMAX = 65504.f;
        const float cx = __builtin_amdgcn_fmed3f(v.x, -FMAX, FMAX), cy = __builtin_amdgcn_fmed3f(v.y, -FMAX, FMAX);
        const float cz = __builtin_amdgcn_fmed3f(v.z, -FMAX, FMAX), cw = __builtin_amdgcn_fmed3f(v.w, -FMAX, FMAX);
        out[0] = make_uint2(pack_f16(cx, cy), pack_f16(cz, cw));
        if constexpr (NP == 2) {
            const unsigned lo = out[0].x, hi = out[0].y;
            out[1] = make_uint2(pack_f16(__builtin_amdgcn_fmed3f(v.x - f16_lo_back(lo), -FMAX, FMAX),
                                         __builtin_amdgcn_fmed3f(v.y - f16_hi_back(lo), -FMAX, FMAX)),
                                pack_f16(__builtin_amdgcn_fmed3f(v.z - f16_lo_back(hi), -FMAX, FMAX),
                                         __builtin_amdgcn_fmed3f(v.w - f16_hi_back(hi), -FMAX, FMAX)));
        }
        return;
    }
    float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const unsigned lo = pack_bf16(r[0], r[1]), hi = pack_bf16(r[2], r[3]);
        out[p] = make_uint2(lo, hi);
        if (p + 1 < NP) {
            r[0] -= __builtin_bit_cast(float, lo << 16);
            r[1] -= __builtin_bit_cast(float, lo & 0xffff0000u);
            r[2] -= __builtin_bit_cast(float, hi << 16);
            r[3] -= __builtin_bit_cast(float, hi & 0xffff0000u);
        }
    }
}

// fp16 range guard (EG3D's modulated_conv2d pre-normalises the styles by their max in its fp16 blocks): the styles
// of the sample are scaled by the power of two 2^-e that brings max|s| into (0.5, 1], so |x * s| <= |x| stays inside
// fp16's range; the accumulators are scaled back by 2^e on the way out.  Powers of two: both scalings are exact, the
// result equals the un-normalised arithmetic.  Every wave reduces the (L2-resident, <= 2 KB) style vector on its own:
// no LDS traffic, no barrier.  Returns 2^-e, *back = 2^e.
// With x_absmax (the producer's max |x| of the whole input tensor, HFAGP_ABSMAX_SLOTS slots) the operand is also
// scaled by the power of two that brings max |x| into [2^14, 2^15): a tensor beyond fp16's range (an fp32 backbone has
// no clamp) cannot saturate, a tiny one keeps all 22 bits of its two parts; again exact.
__device__ __forceinline__ float style_range_guard(const float* styles, int cin, int lane, float* back,
                                                   const float* x_absmax) {
    float m = styles ? 0.f : 1.f;
    if (styles)
        for (int i = lane; i < cin; i += 64) m = fmaxf(m, fabsf(styles[i]));
    float mx = x_absmax ? x_absmax[lane * HFAGP_ABSMAX_STRIDE] : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m = fmaxf(m, __shfl_xor(m, o));
        mx = fmaxf(mx, __shfl_xor(mx, o));
    }
    int e = 0, ex = 15;
    if (m > 0.f && m < 3.0e38f) (void)frexpf(m, &e);
    if (mx > 0.f && mx < 3.0e38f) (void)frexpf(mx, &ex);        // mx = f 2^ex, f in [0.5, 1)
    e += ex - 15;
    e = max(-100, min(100, e));
    *back = ldexpf(1.f, e);
    return ldexpf(1.f, -e);
}

}  // namespace hfagp

// HFAGP_PREC_* -> operand kind of the kernels (0: not a 16-bit precision)
static inline int kind_of(int precision) {
    switch (precision) {
        case HFAGP_PREC_F16: return 1;
        case HFAGP_PREC_BF16X3: return 2;
        case HFAGP_PREC_BF16X6: return 3;
        case HFAGP_PREC_F16X3: return 4;
        case HFAGP_PREC_F16X2: return 5;
        default: return 0;
    }
}

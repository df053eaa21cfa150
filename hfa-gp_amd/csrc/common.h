// Shared helpers for libhfagp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/hfagp.h"

namespace hfagp {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return HFAGP_ELAUNCH;
    }
    return HFAGP_OK;
}

#define HFAGP_REQUIRE(cond, code, ...)            \
    do {                                          \
        if (!(cond)) {                            \
            ::hfagp::set_error(__VA_ARGS__);      \
            return (code);                        \
        }                                         \
    } while (0)

constexpr int kWave = 64;          // gfx950 wavefront
constexpr int kNumCU = 256;        // MI355X
constexpr int kNumXCD = 8;

__device__ __forceinline__ float lrelu_gain_clamp(float v, int act, float alpha, float gain, float clamp) {
    if (act == HFAGP_ACT_LRELU) v = v < 0.f ? v * alpha : v;
    v *= gain;
    if (clamp >= 0.f) v = fminf(fmaxf(v, -clamp), clamp);
    return v;
}

// the two taps (indices into the half-resolution image, weights) of upsample2d (FIR [1,3,3,1], up 2, pad (2,1), gain 4) at
// output coordinate Y; a tap outside the image contributes nothing
__device__ __forceinline__ void up2_taps(int Y, int& i0, int& i1, float& w0, float& w1) {
    if (Y & 1) { i0 = (Y - 1) >> 1; i1 = i0 + 1; w0 = 0.75f; w1 = 0.25f; }
    else       { i1 = Y >> 1; i0 = i1 - 1; w0 = 0.25f; w1 = 0.75f; }
}

// fp16 range tracking (include/hfagp.h, HfagpModconvArgs::y_absmax): wave maximum of |v| -> ONE atomic per wave into
// one of HFAGP_ABSMAX_SLOTS slots (non-negative floats order like their bit patterns; spreading the blocks over the
// slots keeps same-address atomics from serialising in L2).
__device__ __forceinline__ void publish_absmax(float* slots, float m, unsigned slot) {
    // one 128-byte line per slot: 64 slots in two lines queued ~25 k atomics of a 256^2 x 96 tensor on one L2 channel
    // (120 us).  And look before the atomic: a slot only grows, so a (possibly stale, cached) value that already covers
    // this wave's maximum makes the atomic unnecessary.
    unsigned* dst = reinterpret_cast<unsigned*>(slots) + (slot % HFAGP_ABSMAX_SLOTS) * HFAGP_ABSMAX_STRIDE;
    if (__ballot(1) != ~0ull) {                     // a partial wave (tail of a grid): every active lane for itself
        if (__float_as_uint(m) > *dst) atomicMax(dst, __float_as_uint(m));
        return;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && __float_as_uint(m) > *dst) atomicMax(dst, __float_as_uint(m));
}

// XCD-aware bijective remap of a linear block id: blocks that land on the same
// XCD (observed: id % 8) get a contiguous chunk of the logical index space so
// that neighbouring tiles share that XCD's L2 (cdna_hip_programming.md T1).
__device__ __forceinline__ unsigned xcd_remap(unsigned id, unsigned n) {
    const unsigned q = n / kNumXCD, r = n % kNumXCD;
    const unsigned xcd = id % kNumXCD, k = id / kNumXCD;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

}  // namespace hfagp

// HBM-bound streaming ops of the EG3D synthesis path (gfx950).
// Everything here is coalesced 16-byte-per-lane channels-last traffic; nothing is
// reshaped into a GEMM.  See include/hfagp.h for the interfaces they replace.
#include "common.h"

namespace hfagp {

// ---------------------------------------------------------------- error string
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------- styles / demod
// One wave per output element; 64 lanes stride the reduction dimension.
__global__ void __launch_bounds__(256) style_kernel(const float* __restrict__ w, const float* __restrict__ A,
                                                    const float* __restrict__ ab, float* __restrict__ styles,
                                                    int B, int w_dim, int w_stride, int Cin, float wgain, float sgain) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= B * Cin) return;
    const int b = wave / Cin, i = wave % Cin;
    const float* wr = w + (size_t)b * w_stride;
    const float* ar = A + (size_t)i * w_dim;
    float acc = 0.f;
    for (int k = lane * 4; k < w_dim; k += 256) {
        const float4 x = *reinterpret_cast<const float4*>(wr + k);
        const float4 y = *reinterpret_cast<const float4*>(ar + k);
        acc += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) styles[wave] = (acc * wgain + ab[i]) * sgain;
}

__global__ void __launch_bounds__(256) demod_kernel(const float* __restrict__ styles, const float* __restrict__ wsq,
                                                    float* __restrict__ dcoef, int B, int Cin, int Cout, float eps) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= B * Cout) return;
    const int b = wave / Cout, o = wave % Cout;
    const float* s = styles + (size_t)b * Cin;
    const float* q = wsq + (size_t)o * Cin;
    float acc = 0.f;
    for (int k = lane; k < Cin; k += 64) acc += s[k] * s[k] * q[k];
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) dcoef[wave] = rsqrtf(acc + eps);
}

// The same two kernels for up to kStyleBatch layers in ONE launch each (blockIdx.y = layer): a synthesis pass needs
// the styles of 26 layers and the demodulation coefficients of 17; they only depend on ws, so they are computed up
// front in 2 + 2 launches (backbone, super-resolution) instead of 43.
constexpr int kStyleBatch = 32;
struct StyleBatch { HfagpStyleArgs it[kStyleBatch]; };

__global__ void __launch_bounds__(256) style_batch_kernel(const StyleBatch t) {
    const HfagpStyleArgs& a = t.it[blockIdx.y];
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= a.B * a.Cin) return;
    const int b = wave / a.Cin, i = wave % a.Cin;
    const float* wr = a.w + (size_t)b * a.w_stride;
    const float* ar = a.affine_w + (size_t)i * a.w_dim;
    float acc = 0.f;
    for (int k = lane * 4; k < a.w_dim; k += 256) {
        const float4 x = *reinterpret_cast<const float4*>(wr + k);
        const float4 y = *reinterpret_cast<const float4*>(ar + k);
        acc += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) a.styles[wave] = (acc * rsqrtf((float)a.w_dim) + a.affine_b[i]) * a.style_gain;
}

__global__ void __launch_bounds__(256) demod_batch_kernel(const StyleBatch t) {
    const HfagpStyleArgs& a = t.it[blockIdx.y];
    if (!a.dcoef) return;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= a.B * a.Cout) return;
    const int b = wave / a.Cout, o = wave % a.Cout;
    const float* s = a.styles + (size_t)b * a.Cin;
    const float* q = a.wsq + (size_t)o * a.Cin;
    float acc = 0.f;
    for (int k = lane; k < a.Cin; k += 64) acc += s[k] * s[k] * q[k];
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) a.dcoef[wave] = rsqrtf(acc + a.eps);
}

// ---------------------------------------------------------------- fully connected (mapping network)
// y[b][o] = act( (x[b] . W[o]) * wgain + bias[o] * bgain ) * act_gain ;  one wave per output element
__global__ void __launch_bounds__(256) fc_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                 const float* __restrict__ bias, float* __restrict__ y, int B, int In,
                                                 int Out, float wgain, float bgain, int act, float alpha, float gain) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= B * Out) return;
    const int b = wave / Out, o = wave % Out;
    float acc = 0.f;
    for (int k = lane; k < In; k += 64) acc += x[(size_t)b * In + k] * W[(size_t)o * In + k];
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) y[wave] = lrelu_gain_clamp(acc * wgain + (bias ? bias[o] * bgain : 0.f), act, alpha, gain, -1.f);
}

// ---------------------------------------------------------------- weight prep
__global__ void __launch_bounds__(256) weight_prep_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                          float* __restrict__ wsq, int Cout, int Cin, int taps) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over Cout*Cin
    if (idx >= Cout * Cin) return;
    const int co = idx / Cin, ci = idx % Cin;
    const float* src = w + (size_t)idx * taps;
    float sq = 0.f;
    for (int t = 0; t < taps; ++t) {
        const float v = src[t];
        sq += v * v;
        wt[(((size_t)t * (Cin >> 2) + (ci >> 2)) * Cout + co) * 4 + (ci & 3)] = v;
    }
    if (wsq) wsq[idx] = sq;
}

// ---------------------------------------------------------------- FIR + epilogue after the transposed conv
// yt [B][2H+1][2W+1][C] -> y [B][2H][2W][C];  out = sum_{p,q} f[p] f[q] yt[Y+p-1][X+q-1],
// f = [1,3,3,1]/4 per axis (= outer([1,3,3,1])/64 * gain 4).  Each thread owns 4 channels
// of one output column and walks a vertical strip with a sliding window of
// horizontally filtered rows, so every input row is read once per column.
// Each thread owns 4 channels of TWO adjacent output columns and walks a vertical strip: 5 input columns per
// row feed both columns (2.5 loads per output row instead of 4), every input row of the strip is read once, and
// the image borders are handled by clamped addresses with zeroed tap weights (no branches in the loop).
#ifndef HFAGP_FIR_STRIP
#define HFAGP_FIR_STRIP 8
#endif
constexpr int kStrip = HFAGP_FIR_STRIP;     // output rows per thread of upfir_epilogue_kernel

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
// 4 channels of one pixel: fp32 (float4) or fp16 storage (8 bytes) — io_f16 of HfagpUpfirEpilogueArgs / HfagpModconvArgs
template <bool H16>
__device__ __forceinline__ float4 load4(const void* base, size_t idx4) {
    if constexpr (H16) {
        const uint2 u = reinterpret_cast<const uint2*>(base)[idx4];
        const h2_t a = __builtin_bit_cast(h2_t, u.x), b = __builtin_bit_cast(h2_t, u.y);
        return make_float4((float)a[0], (float)a[1], (float)b[0], (float)b[1]);
    } else {
        return reinterpret_cast<const float4*>(base)[idx4];
    }
}
template <bool H16>
__device__ __forceinline__ void store4(void* base, size_t idx4, float4 v) {
    if constexpr (H16) {
        const f2_t lo = {v.x, v.y}, hi = {v.z, v.w};
        reinterpret_cast<uint2*>(base)[idx4] = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(lo, h2_t)),
                                                          __builtin_bit_cast(unsigned, __builtin_convertvector(hi, h2_t)));
    } else {
        reinterpret_cast<float4*>(base)[idx4] = v;
    }
}

template <bool H16>
__global__ void __launch_bounds__(256) upfir_epilogue_kernel(HfagpUpfirEpilogueArgs a) {
    const int C4 = a.C >> 2;
    const int Wo = 2 * a.W, Ho = 2 * a.H, Wi = 2 * a.W + 1, Hi = 2 * a.H + 1;
    const int Wp = a.W;                                      // column pairs
    const int strips = (Ho + kStrip - 1) / kStrip;
    const long long total = (long long)a.B * strips * Wp * C4;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= total) return;
    const int c4 = (int)(tid % C4);
    const int X = 2 * (int)((tid / C4) % Wp);
    const int st = (int)((tid / ((long long)C4 * Wp)) % strips);
    const int b = (int)(tid / ((long long)C4 * Wp * strips));
    const int Y0 = st * kStrip;
    const size_t src0 = (size_t)b * Hi * Wi * C4 + c4;           // in units of 4 channels
    const float f0 = 0.25f, f1 = 0.75f;

    // input columns X-1 .. X+3: clamped offset + validity
    int xo[5]; float xm[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int xin = X + q - 1;
        xm[q] = (xin >= 0 && xin < Wi) ? 1.f : 0.f;
        xo[q] = min(max(xin, 0), Wi - 1) * C4;
    }
    const float wa[4] = {f0 * xm[0], f1 * xm[1], f1 * xm[2], f0 * xm[3]};     // output column X
    const float wb[4] = {f0 * xm[1], f1 * xm[2], f1 * xm[3], f0 * xm[4]};     // output column X+1

    auto hrow = [&](int yin, float4& ha, float4& hb) {
        const float m = (yin >= 0 && yin < Hi) ? 1.f : 0.f;
        const size_t row = src0 + (size_t)min(max(yin, 0), Hi - 1) * Wi * C4;
        float4 v[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) v[q] = load4<H16>(a.yt, row + xo[q]);
        ha.x = m * (wa[0] * v[0].x + wa[1] * v[1].x + wa[2] * v[2].x + wa[3] * v[3].x);
        ha.y = m * (wa[0] * v[0].y + wa[1] * v[1].y + wa[2] * v[2].y + wa[3] * v[3].y);
        ha.z = m * (wa[0] * v[0].z + wa[1] * v[1].z + wa[2] * v[2].z + wa[3] * v[3].z);
        ha.w = m * (wa[0] * v[0].w + wa[1] * v[1].w + wa[2] * v[2].w + wa[3] * v[3].w);
        hb.x = m * (wb[0] * v[1].x + wb[1] * v[2].x + wb[2] * v[3].x + wb[3] * v[4].x);
        hb.y = m * (wb[0] * v[1].y + wb[1] * v[2].y + wb[2] * v[3].y + wb[3] * v[4].y);
        hb.z = m * (wb[0] * v[1].z + wb[1] * v[2].z + wb[2] * v[3].z + wb[3] * v[4].z);
        hb.w = m * (wb[0] * v[1].w + wb[1] * v[2].w + wb[2] * v[3].w + wb[3] * v[4].w);
    };

    float4 d = make_float4(1.f, 1.f, 1.f, 1.f), bs = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.dcoef) d = reinterpret_cast<const float4*>(a.dcoef + (size_t)b * a.C)[c4];
    if (a.bias) bs = reinterpret_cast<const float4*>(a.bias)[c4];

    float4 a0, a1, a2, b0, b1, b2;
    hrow(Y0 - 1, a0, b0); hrow(Y0, a1, b1); hrow(Y0 + 1, a2, b2);
    const size_t dst0 = (size_t)b * Ho * Wo * C4 + c4;
    float vmax = 0.f;                                        // max |y| of this thread's stores (a.y_absmax)
    auto finish = [&](float4 o, float nz) -> float4 {
        o.x = lrelu_gain_clamp(o.x * d.x + nz + bs.x, a.act, a.alpha, a.gain, a.clamp);
        o.y = lrelu_gain_clamp(o.y * d.y + nz + bs.y, a.act, a.alpha, a.gain, a.clamp);
        o.z = lrelu_gain_clamp(o.z * d.z + nz + bs.z, a.act, a.alpha, a.gain, a.clamp);
        o.w = lrelu_gain_clamp(o.w * d.w + nz + bs.w, a.act, a.alpha, a.gain, a.clamp);
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        return o;
    };
#pragma unroll
    for (int k = 0; k < kStrip; ++k) {
        const int Y = Y0 + k;
        if (Y >= Ho) break;
        float4 a3, b3;
        hrow(Y + 2, a3, b3);
        float4 oa, ob;
        oa.x = f0 * a0.x + f1 * a1.x + f1 * a2.x + f0 * a3.x;
        oa.y = f0 * a0.y + f1 * a1.y + f1 * a2.y + f0 * a3.y;
        oa.z = f0 * a0.z + f1 * a1.z + f1 * a2.z + f0 * a3.z;
        oa.w = f0 * a0.w + f1 * a1.w + f1 * a2.w + f0 * a3.w;
        ob.x = f0 * b0.x + f1 * b1.x + f1 * b2.x + f0 * b3.x;
        ob.y = f0 * b0.y + f1 * b1.y + f1 * b2.y + f0 * b3.y;
        ob.z = f0 * b0.z + f1 * b1.z + f1 * b2.z + f0 * b3.z;
        ob.w = f0 * b0.w + f1 * b1.w + f1 * b2.w + f0 * b3.w;
        float nza = 0.f, nzb = 0.f;
        if (a.noise) {
            nza = a.noise[(size_t)Y * Wo + X] * a.noise_strength;
            nzb = a.noise[(size_t)Y * Wo + X + 1] * a.noise_strength;
        }
        store4<H16>(a.y, dst0 + ((size_t)Y * Wo + X) * C4, finish(oa, nza));
        store4<H16>(a.y, dst0 + ((size_t)Y * Wo + X + 1) * C4, finish(ob, nzb));
        a0 = a1; a1 = a2; a2 = a3;
        b0 = b1; b1 = b2; b2 = b3;
    }
    if (a.y_absmax) publish_absmax(a.y_absmax, vmax, blockIdx.x * 4 + (threadIdx.x >> 6));
}

// ---------------------------------------------------------------- skip: img_out = upsample2d(img_in) + y
// upsample2d = zero-insert x2, pad [2,1,2,1], FIR [1,3,3,1]^2/64, gain 4  ==  per axis
//   out[2i]   = .25*in[i-1] + .75*in[i]
//   out[2i+1] = .75*in[i]   + .25*in[i+1]          (out-of-range taps are zero)

__global__ void __launch_bounds__(256) skip_kernel(HfagpSkipArgs a) {
    const int C4 = a.C >> 2;
    const int Ho = a.img_in ? 2 * a.H : a.H, Wo = a.img_in ? 2 * a.W : a.W;
    const long long total = (long long)a.B * Ho * Wo * C4;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= total) return;
    const int c4 = (int)(tid % C4);
    const int X = (int)((tid / C4) % Wo);
    const int Y = (int)((tid / ((long long)C4 * Wo)) % Ho);
    const int b = (int)(tid / ((long long)C4 * Wo * Ho));
    float4 o = reinterpret_cast<const float4*>(a.y)[tid];
    if (a.img_in) {
        int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
        up2_taps(Y, y0, y1, wy0, wy1);
        up2_taps(X, x0, x1, wx0, wx1);
        const float4* src = reinterpret_cast<const float4*>(a.img_in) + (size_t)b * a.H * a.W * C4 + c4;
        const int ys[2] = {y0, y1}, xs[2] = {x0, x1};
        const float wy[2] = {wy0, wy1}, wx[2] = {wx0, wx1};
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (ys[p] >= 0 && ys[p] < a.H && xs[q] >= 0 && xs[q] < a.W) {
                    const float4 v = src[((size_t)ys[p] * a.W + xs[q]) * C4];
                    const float wgt = wy[p] * wx[q];
                    o.x = fmaf(wgt, v.x, o.x); o.y = fmaf(wgt, v.y, o.y); o.z = fmaf(wgt, v.z, o.z); o.w = fmaf(wgt, v.w, o.w);
                }
    }
    if (a.plane_major) {
        const int Cp = a.C / 3;                     // channels per plane
        const int c = c4 * 4, pl = c / Cp, ch = c % Cp;
        float4* dst = reinterpret_cast<float4*>(a.img_out + ((((size_t)b * 3 + pl) * Ho + Y) * Wo + X) * Cp + ch);
        *dst = o;
    } else {
        reinterpret_cast<float4*>(a.img_out)[tid] = o;
    }
    if (a.out_absmax)
        publish_absmax(a.out_absmax, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))),
                       blockIdx.x * 4 + (threadIdx.x >> 6));
}

// ---------------------------------------------------------------- Blur(pad 1) + stride-2 sampling, channels-last
// (front of the 1x1 skip conv of the RGB driver's ResBlock) and its adjoint; thread = (pixel, 4 channels)
__global__ void __launch_bounds__(256) blur_down_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H,
                                                        int W, int C4) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long long total = (long long)B * Ho * Wo * C4;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= total) return;
    const int c4 = (int)(tid % C4);
    const int j = (int)((tid / C4) % Wo);
    const int i = (int)((tid / ((long long)C4 * Wo)) % Ho);
    const int b = (int)(tid / ((long long)C4 * Wo * Ho));
    const float f[4] = {0.125f, 0.375f, 0.375f, 0.125f};
    const float4* src = reinterpret_cast<const float4*>(x) + (size_t)b * H * W * C4 + c4;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = 2 * i + a - 1;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int xx = 2 * j + q - 1;
            if (xx < 0 || xx >= W) continue;
            const float4 v = src[((size_t)yy * W + xx) * C4];
            const float wgt = f[a] * f[q];
            o.x = fmaf(wgt, v.x, o.x); o.y = fmaf(wgt, v.y, o.y); o.z = fmaf(wgt, v.z, o.z); o.w = fmaf(wgt, v.w, o.w);
        }
    }
    reinterpret_cast<float4*>(y)[tid] = o;
}

// gx[y][x] = sum over (i, a): 2i + a - 1 = y, (j, q): 2j + q - 1 = x of f[a] f[q] gy[i][j]  (two candidates per axis)
__global__ void __launch_bounds__(256) blur_down_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int B,
                                                            int H, int W, int C4) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long long total = (long long)B * H * W * C4;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= total) return;
    const int c4 = (int)(tid % C4);
    const int X = (int)((tid / C4) % W);
    const int Y = (int)((tid / ((long long)C4 * W)) % H);
    const int b = (int)(tid / ((long long)C4 * W * H));
    const float f[4] = {0.125f, 0.375f, 0.375f, 0.125f};
    // a = Y + 1 - 2i in [0, 4): i = (Y + 1 - a) / 2 for the two a of the right parity
    const int a0 = (Y + 1) & 1, q0 = (X + 1) & 1;
    const float4* src = reinterpret_cast<const float4*>(gy) + (size_t)b * Ho * Wo * C4 + c4;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int a = a0 + 2 * u, i = (Y + 1 - a) >> 1;
        if (Y + 1 - a < 0 || i >= Ho) continue;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int q = q0 + 2 * v, j = (X + 1 - q) >> 1;
            if (X + 1 - q < 0 || j >= Wo) continue;
            const float4 g = src[((size_t)i * Wo + j) * C4];
            const float wgt = f[a] * f[q];
            o.x = fmaf(wgt, g.x, o.x); o.y = fmaf(wgt, g.y, o.y); o.z = fmaf(wgt, g.z, o.z); o.w = fmaf(wgt, g.w, o.w);
        }
    }
    reinterpret_cast<float4*>(gx)[tid] = o;
}

// ---------------------------------------------------------------- toRGB (few output channels) + skip
// 8 lanes per pixel, each lane strides the input channels in float4 steps, so a wave
// reads 8 full 128-byte-aligned runs per load instruction.
constexpr int kTorgbMaxOut = 4;

// epilogue of one (pixel, output channel c): bias, (pre-clamp copy), clamp, + upsample2d(rgb_in), store.
// The 8 lanes of a pixel all hold the reduced sums; lane c finishes channel c, so the 4-tap skip gathers of the
// channels run side by side instead of one lane walking all of them.
// upsample2d(rgb_in) at output pixel `pix`, channel c (0 when there is no previous image): four unconditional loads
// (clamped addresses + zeroed weights).
__device__ __forceinline__ float torgb_skip(const HfagpTorgbArgs& a, int b, int pix, int c) {
    if (!a.rgb_in) return 0.f;
    const int Y = pix / a.W, X = pix % a.W;
    const int Hi = a.H >> 1, Wi = a.W >> 1;
    int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
    up2_taps(Y, y0, y1, wy0, wy1);
    up2_taps(X, x0, x1, wx0, wx1);
    const float* src = a.rgb_in + ((size_t)b * a.Cout + c) * Hi * Wi;
    const float my0 = y0 >= 0 ? wy0 : 0.f, my1 = y1 < Hi ? wy1 : 0.f;
    const float mx0 = x0 >= 0 ? wx0 : 0.f, mx1 = x1 < Wi ? wx1 : 0.f;
    const int cy0 = max(y0, 0), cy1 = min(y1, Hi - 1), cx0 = max(x0, 0), cx1 = min(x1, Wi - 1);
    return my0 * (mx0 * src[cy0 * Wi + cx0] + mx1 * src[cy0 * Wi + cx1]) +
           my1 * (mx0 * src[cy1 * Wi + cx0] + mx1 * src[cy1 * Wi + cx1]);
}

__device__ __forceinline__ void torgb_finish(const HfagpTorgbArgs& a, int b, int pix, int c, float acc, float skip) {
    const int HW = a.H * a.W;
    float v = acc + a.bias[c];
    if (a.y_pre) a.y_pre[((size_t)b * a.Cout + c) * HW + pix] = v;      // before the clamp: backward mask
    if (a.clamp >= 0.f) v = fminf(fmaxf(v, -a.clamp), a.clamp);
    a.rgb_out[((size_t)b * a.Cout + c) * HW + pix] = v + skip;
}

// Cin = 32*KQ: lane `sub` of a pixel owns the channel quads sub, sub+8, ...; its slice of the modulated weight
// stays in registers and the block walks kTorgbPix pixels, so the KQ loads of a pixel are all in flight together
// and the weight set-up is paid once per 32*kTorgbIter pixels.
constexpr int kTorgbIter = 8;

// NO = number of output accumulators kept per lane (3 for RGB: a quarter fewer weight registers and FMAs than 4)
template <int KQ, int NO>
__global__ void __launch_bounds__(256) torgb_reg_kernel(HfagpTorgbArgs a) {
    const int b = blockIdx.y;
    const int sub = threadIdx.x & 7;
    const int HW = a.H * a.W;
    float4 w[NO][KQ];
#pragma unroll
    for (int c = 0; c < NO; ++c)
#pragma unroll
        for (int i = 0; i < KQ; ++i) {
            w[c][i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < a.Cout) {
                const float4 wv = reinterpret_cast<const float4*>(a.weight + (size_t)c * a.Cin)[sub + 8 * i];
                const float4 sv = reinterpret_cast<const float4*>(a.styles + (size_t)b * a.Cin)[sub + 8 * i];
                w[c][i] = make_float4(wv.x * sv.x, wv.y * sv.y, wv.z * sv.z, wv.w * sv.w);
            }
        }
    const int pix0 = blockIdx.x * (32 * kTorgbIter) + (threadIdx.x >> 3);
    const float4* xb = reinterpret_cast<const float4*>(a.x + (size_t)b * HW * a.Cin) + sub;
    const int cq = a.Cin >> 2;
    float4 v[KQ], vn[KQ];
#pragma unroll
    for (int i = 0; i < KQ; ++i) v[i] = xb[(size_t)min(pix0, HW - 1) * cq + 8 * i];
    for (int it = 0; it < kTorgbIter; ++it) {
        const int pix = pix0 + it * 32;
        if (pix >= HW) return;
        // next pixel's loads go out before this pixel's reduction and epilogue
        const int pn = min(pix + 32, HW - 1);
#pragma unroll
        for (int i = 0; i < KQ; ++i) vn[i] = xb[(size_t)pn * cq + 8 * i];
        float acc[NO];
#pragma unroll
        for (int c = 0; c < NO; ++c) acc[c] = 0.f;
#pragma unroll
        for (int i = 0; i < KQ; ++i)
#pragma unroll
            for (int c = 0; c < NO; ++c)
                acc[c] += v[i].x * w[c][i].x + v[i].y * w[c][i].y + v[i].z * w[c][i].z + v[i].w * w[c][i].w;
#pragma unroll
        for (int c = 0; c < NO; ++c) {
            acc[c] += __shfl_xor(acc[c], 1);
            acc[c] += __shfl_xor(acc[c], 2);
            acc[c] += __shfl_xor(acc[c], 4);
        }
        if (sub < a.Cout) {
            float mine = acc[0];
#pragma unroll
            for (int c = 1; c < NO; ++c) mine = sub == c ? acc[c] : mine;
            torgb_finish(a, b, pix, sub, mine, torgb_skip(a, b, pix, sub));   // (issuing the skip loads before the
                                                                              // reduction was measured: slower)
        }
#pragma unroll
        for (int i = 0; i < KQ; ++i) v[i] = vn[i];
    }
}

// any Cin % 4 == 0: modulated weight in LDS
__global__ void __launch_bounds__(256) torgb_kernel(HfagpTorgbArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wmod[];   // [Cout][Cin]
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < a.Cout * a.Cin; i += blockDim.x)
        wmod[i] = a.weight[i] * a.styles[(size_t)b * a.Cin + (i % a.Cin)];
    __syncthreads();
    const int HW = a.H * a.W;
    const int sub = threadIdx.x & 7;
    const int pix = blockIdx.x * (blockDim.x >> 3) + (threadIdx.x >> 3);
    if (pix >= HW) return;
    const float4* xr = reinterpret_cast<const float4*>(a.x + ((size_t)b * HW + pix) * a.Cin);
    float acc[kTorgbMaxOut] = {0.f, 0.f, 0.f, 0.f};
    for (int k = sub; k < (a.Cin >> 2); k += 8) {
        const float4 v = xr[k];
#pragma unroll
        for (int c = 0; c < kTorgbMaxOut; ++c)
            if (c < a.Cout) {
                const float4 wv = reinterpret_cast<const float4*>(wmod + c * a.Cin)[k];
                acc[c] += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
            }
    }
#pragma unroll
    for (int c = 0; c < kTorgbMaxOut; ++c) {
        acc[c] += __shfl_xor(acc[c], 1);
        acc[c] += __shfl_xor(acc[c], 2);
        acc[c] += __shfl_xor(acc[c], 4);
    }
    if (sub < a.Cout)
        torgb_finish(a, b, pix, sub, sub == 0 ? acc[0] : sub == 1 ? acc[1] : sub == 2 ? acc[2] : acc[3], torgb_skip(a, b, pix, sub));
}

// second half of the fused toRGB: one thread per pixel sums the partial images, then bias / clamp / skip per channel
__global__ void __launch_bounds__(256) torgb_finish_kernel(HfagpTorgbFinishArgs f) {
    const int HW = f.H * f.W;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long long)f.B * HW) return;
    const int b = (int)(tid / HW), pix = (int)(tid % HW);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q = 0; q < f.nparts; ++q) {
        const float4 v = reinterpret_cast<const float4*>(f.part)[((size_t)q * f.B + b) * HW + pix];
        acc.x += v.x; acc.y += v.y; acc.z += v.z;
    }
    HfagpTorgbArgs a{};                       // torgb_skip / torgb_finish read these fields only
    a.bias = f.bias; a.rgb_in = f.rgb_in; a.rgb_out = f.rgb_out; a.y_pre = f.y_pre;
    a.H = f.H; a.W = f.W; a.Cout = f.Cout; a.clamp = f.clamp;
    const float s3[3] = {acc.x, acc.y, acc.z};
#pragma unroll
    for (int c = 0; c < 3; ++c)
        if (c < f.Cout) torgb_finish(a, b, pix, c, s3[c], torgb_skip(a, b, pix, c));
}

// ---------------------------------------------------------------- generic upfirdn2d (NCHW, test surface)
__global__ void __launch_bounds__(256) upfirdn2d_kernel(const float* __restrict__ x, const float* __restrict__ f,
                                                        float* __restrict__ y, int NC, int H, int W, int fh, int fw,
                                                        int up, int down, int px0, int py0, int Ho, int Wo, float gain,
                                                        int noflip) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long long)NC * Ho * Wo) return;
    const int ox = (int)(tid % Wo), oy = (int)((tid / Wo) % Ho);
    const int nc = (int)(tid / ((long long)Wo * Ho));
    const float* src = x + (size_t)nc * H * W;
    float acc = 0.f;
    // padded/upsampled coordinate of the window origin; correlate with the flipped filter
    for (int p = 0; p < fh; ++p) {
        const int uy = oy * down + p - py0;
        if (uy < 0 || uy % up != 0) continue;
        const int iy = uy / up;
        if (iy >= H) continue;
        for (int q = 0; q < fw; ++q) {
            const int ux = ox * down + q - px0;
            if (ux < 0 || ux % up != 0) continue;
            const int ix = ux / up;
            if (ix >= W) continue;
            // forward: true convolution (filter flipped, EG3D flip_filter=False); the adjoint correlates (noflip)
            acc += (noflip ? f[p * fw + q] : f[(fh - 1 - p) * fw + (fw - 1 - q)]) * src[(size_t)iy * W + ix];
        }
    }
    y[tid] = acc * gain;
}

// ---------------------------------------------------------------- bias_act (test surface)
__global__ void __launch_bounds__(256) bias_act_kernel(const float* __restrict__ x, const float* __restrict__ b,
                                                       float* __restrict__ y, long long n, int C, long long inner,
                                                       int act, float alpha, float gain, float clamp) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = x[i];
        if (b) v += b[(i / inner) % C];
        y[i] = lrelu_gain_clamp(v, act, alpha, gain, clamp);
    }
}

// adjoint of bias_act w.r.t. x (and, summed over everything but the channel, b): EG3D's bias_act backward works from the
// OUTPUT y (sign(y) = sign(x + b) for leaky ReLU; the clamp passes gradient where |y| < clamp)
__global__ void __launch_bounds__(256) bias_act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                           float* __restrict__ dx, long long n, int act, float alpha,
                                                           float gain, float clamp) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float yv = y[i];
        float g = dy[i] * gain;
        if (act == HFAGP_ACT_LRELU && !(yv > 0.f)) g *= alpha;     // slope alpha AT 0 too: ATen's leaky_relu_backward, EG3D's bias_act
        if (clamp >= 0.f && !(fabsf(yv) < clamp)) g = 0.f;
        dx[i] = g;
    }
}

// ---------------------------------------------------------------- layout transposes (32x33 LDS tile)
__global__ void __launch_bounds__(256) transpose_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        int rows, int cols) {
    // x [batch][rows][cols] -> y [batch][cols][rows]
    __shared__ float tile[32][33];
    const size_t base = (size_t)blockIdx.z * rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        if (r < rows && c < cols) tile[k][tx] = x[base + (size_t)r * cols + c];
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (r < rows && c < cols) y[base + (size_t)c * rows + r] = tile[tx][k];
    }
}

}  // namespace hfagp

using namespace hfagp;

extern "C" {

int hfagp_abi_version(void) { return HFAGP_ABI_VERSION; }
const char* hfagp_last_error(void) { return g_err; }

int hfagp_style_fwd(const HfagpStyleArgs* a, void* stream) {
    HFAGP_REQUIRE(a && a->w && a->affine_w && a->affine_b && a->styles, HFAGP_EBADARG, "style_fwd: null pointer");
    HFAGP_REQUIRE(a->B > 0 && a->Cin > 0 && a->w_dim > 0 && a->w_dim % 4 == 0 && a->w_stride % 4 == 0,
                  HFAGP_EBADARG, "style_fwd: bad dims B=%d Cin=%d w_dim=%d", a->B, a->Cin, a->w_dim);
    hipStream_t s = (hipStream_t)stream;
    const float wgain = 1.0f / sqrtf((float)a->w_dim);
    int waves = a->B * a->Cin;
    style_kernel<<<(waves + 3) / 4, 256, 0, s>>>(a->w, a->affine_w, a->affine_b, a->styles, a->B, a->w_dim,
                                                 a->w_stride, a->Cin, wgain, a->style_gain);
    if (a->dcoef) {
        HFAGP_REQUIRE(a->wsq && a->Cout > 0, HFAGP_EBADARG, "style_fwd: dcoef requested without wsq");
        waves = a->B * a->Cout;
        demod_kernel<<<(waves + 3) / 4, 256, 0, s>>>(a->styles, a->wsq, a->dcoef, a->B, a->Cin, a->Cout, a->eps);
    }
    return check_launch("style_fwd");
}

int hfagp_style_batch_fwd(const HfagpStyleArgs* items, int32_t n, void* stream) {
    HFAGP_REQUIRE(items && n >= 1 && n <= kStyleBatch, HFAGP_EBADARG, "style_batch_fwd: 1..%d items, got %d", kStyleBatch, n);
    StyleBatch t;
    int max_s = 0, max_d = 0;
    for (int i = 0; i < n; ++i) {
        const HfagpStyleArgs& a = items[i];
        HFAGP_REQUIRE(a.w && a.affine_w && a.affine_b && a.styles, HFAGP_EBADARG, "style_batch_fwd: null pointer in item %d", i);
        HFAGP_REQUIRE(a.B > 0 && a.Cin > 0 && a.w_dim > 0 && a.w_dim % 4 == 0 && a.w_stride % 4 == 0, HFAGP_EBADARG,
                      "style_batch_fwd: bad dims in item %d: B=%d Cin=%d w_dim=%d", i, a.B, a.Cin, a.w_dim);
        HFAGP_REQUIRE(!a.dcoef || (a.wsq && a.Cout > 0), HFAGP_EBADARG, "style_batch_fwd: item %d: dcoef without wsq", i);
        t.it[i] = a;
        max_s = a.B * a.Cin > max_s ? a.B * a.Cin : max_s;
        if (a.dcoef) max_d = a.B * a.Cout > max_d ? a.B * a.Cout : max_d;
    }
    hipStream_t s = (hipStream_t)stream;
    style_batch_kernel<<<dim3((max_s + 3) / 4, n), 256, 0, s>>>(t);
    if (max_d) demod_batch_kernel<<<dim3((max_d + 3) / 4, n), 256, 0, s>>>(t);
    return check_launch("style_batch_fwd");
}

int hfagp_fc_fwd(const float* x, const float* weight, const float* bias, float* y, int32_t B, int32_t In, int32_t Out,
                 float lr_mul, int32_t act, float alpha, float gain, void* stream) {
    HFAGP_REQUIRE(x && weight && y, HFAGP_EBADARG, "fc_fwd: null pointer");
    HFAGP_REQUIRE(B > 0 && In > 0 && Out > 0, HFAGP_EBADARG, "fc_fwd: bad dims");
    HFAGP_REQUIRE(act == HFAGP_ACT_LINEAR || act == HFAGP_ACT_LRELU, HFAGP_EUNSUPPORTED, "fc_fwd: act %d", act);
    const int waves = B * Out;
    fc_kernel<<<(waves + 3) / 4, 256, 0, (hipStream_t)stream>>>(x, weight, bias, y, B, In, Out,
                                                                lr_mul / sqrtf((float)In), lr_mul, act, alpha, gain);
    return check_launch("fc_fwd");
}

int hfagp_weight_prep(const float* weight, float* wt, float* wsq, int32_t Cout, int32_t Cin, int32_t taps,
                      void* stream) {
    HFAGP_REQUIRE(weight && wt, HFAGP_EBADARG, "weight_prep: null pointer");
    HFAGP_REQUIRE(Cin % 4 == 0 && Cout > 0 && (taps == 1 || taps == 9), HFAGP_EUNSUPPORTED,
                  "weight_prep: Cin=%d must be a multiple of 4, taps=%d must be 1 or 9", Cin, taps);
    const int n = Cout * Cin;
    weight_prep_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(weight, wt, wsq, Cout, Cin, taps);
    return check_launch("weight_prep");
}

int hfagp_upfir_epilogue_fwd(const HfagpUpfirEpilogueArgs* a, void* stream) {
    HFAGP_REQUIRE(a && a->yt && a->y, HFAGP_EBADARG, "upfir_epilogue: null pointer");
    HFAGP_REQUIRE(a->C % 4 == 0 && a->B > 0 && a->H > 0 && a->W > 0, HFAGP_EUNSUPPORTED,
                  "upfir_epilogue: C=%d must be a multiple of 4", a->C);
    const int strips = (2 * a->H + kStrip - 1) / kStrip;
    const long long total = (long long)a->B * strips * a->W * (a->C / 4);
    if (a->io_f16) upfir_epilogue_kernel<true><<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(*a);
    else upfir_epilogue_kernel<false><<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(*a);
    return check_launch("upfir_epilogue");
}

int hfagp_skip_upsample_add(const HfagpSkipArgs* a, void* stream) {
    HFAGP_REQUIRE(a && a->y && a->img_out, HFAGP_EBADARG, "skip_upsample_add: null pointer");
    HFAGP_REQUIRE(a->C % 4 == 0 && (!a->plane_major || (a->C % 3 == 0 && (a->C / 3) % 4 == 0)), HFAGP_EUNSUPPORTED,
                  "skip_upsample_add: unsupported channel count %d", a->C);
    const int Ho = a->img_in ? 2 * a->H : a->H, Wo = a->img_in ? 2 * a->W : a->W;
    const long long total = (long long)a->B * Ho * Wo * (a->C / 4);
    skip_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(*a);
    return check_launch("skip_upsample_add");
}

int hfagp_torgb_fwd(const HfagpTorgbArgs* a, void* stream) {
    HFAGP_REQUIRE(a && a->x && a->weight && a->styles && a->bias && a->rgb_out, HFAGP_EBADARG, "torgb: null pointer");
    HFAGP_REQUIRE(a->Cout >= 1 && a->Cout <= kTorgbMaxOut && a->Cin % 4 == 0 && a->H % 2 == 0 && a->W % 2 == 0,
                  HFAGP_EUNSUPPORTED, "torgb: Cout=%d (max %d), Cin=%d", a->Cout, kTorgbMaxOut, a->Cin);
    const int HW = a->H * a->W;
    hipStream_t s = (hipStream_t)stream;
    const dim3 rgrid((HW + 32 * kTorgbIter - 1) / (32 * kTorgbIter), a->B);
    switch (a->Cin) {
        case 64:  if (a->Cout <= 3) torgb_reg_kernel<2, 3><<<rgrid, 256, 0, s>>>(*a); else torgb_reg_kernel<2, 4><<<rgrid, 256, 0, s>>>(*a); break;
        case 128: if (a->Cout <= 3) torgb_reg_kernel<4, 3><<<rgrid, 256, 0, s>>>(*a); else torgb_reg_kernel<4, 4><<<rgrid, 256, 0, s>>>(*a); break;
        case 256: if (a->Cout <= 3) torgb_reg_kernel<8, 3><<<rgrid, 256, 0, s>>>(*a); else torgb_reg_kernel<8, 4><<<rgrid, 256, 0, s>>>(*a); break;
        default: {
            dim3 grid((HW + 31) / 32, a->B);
            const size_t lds = (size_t)a->Cout * a->Cin * sizeof(float);
            torgb_kernel<<<grid, 256, lds, s>>>(*a);
        }
    }
    return check_launch("torgb");
}

int hfagp_torgb_finish_fwd(const HfagpTorgbFinishArgs* a, void* stream) {
    HFAGP_REQUIRE(a && a->part && a->bias && a->rgb_out, HFAGP_EBADARG, "torgb_finish: null pointer");
    HFAGP_REQUIRE(a->Cout >= 1 && a->Cout <= 3 && a->nparts >= 1 && a->H % 2 == 0 && a->W % 2 == 0, HFAGP_EUNSUPPORTED,
                  "torgb_finish: Cout=%d (max 3), nparts=%d", a->Cout, a->nparts);
    const long long total = (long long)a->B * a->H * a->W;
    torgb_finish_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(*a);
    return check_launch("torgb_finish");
}

int hfagp_blur_down_fwd(const float* x, float* y, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
    HFAGP_REQUIRE(x && y, HFAGP_EBADARG, "blur_down: null pointer");
    HFAGP_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, HFAGP_EUNSUPPORTED,
                  "blur_down: H=%d, W=%d must be even and C=%d a multiple of 4", H, W, C);
    const long long total = (long long)B * (H / 2) * (W / 2) * (C / 4);
    blur_down_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, y, B, H, W, C / 4);
    return check_launch("blur_down_fwd");
}

int hfagp_blur_down_bwd(const float* gy, float* gx, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
    HFAGP_REQUIRE(gy && gx, HFAGP_EBADARG, "blur_down_bwd: null pointer");
    HFAGP_REQUIRE(B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0, HFAGP_EUNSUPPORTED,
                  "blur_down_bwd: H=%d, W=%d must be even and C=%d a multiple of 4", H, W, C);
    const long long total = (long long)B * H * W * (C / 4);
    blur_down_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(gy, gx, B, H, W, C / 4);
    return check_launch("blur_down_bwd");
}

int hfagp_upfirdn2d_fwd(const float* x, const float* f, float* y, int32_t N, int32_t C, int32_t H, int32_t W,
                        int32_t fh, int32_t fw, int32_t up, int32_t down, int32_t px0, int32_t px1, int32_t py0,
                        int32_t py1, float gain, void* stream) {
    HFAGP_REQUIRE(x && f && y, HFAGP_EBADARG, "upfirdn2d: null pointer");
    HFAGP_REQUIRE(up >= 1 && down >= 1 && fh >= 1 && fw >= 1, HFAGP_EBADARG, "upfirdn2d: bad up/down/filter");
    const int Ho = (H * up + py0 + py1 - fh) / down + 1, Wo = (W * up + px0 + px1 - fw) / down + 1;
    HFAGP_REQUIRE(Ho > 0 && Wo > 0, HFAGP_EBADARG, "upfirdn2d: empty output");
    const long long total = (long long)N * C * Ho * Wo;
    upfirdn2d_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        x, f, y, N * C, H, W, fh, fw, up, down, px0, py0, Ho, Wo, gain, 0);
    return check_launch("upfirdn2d");
}

int hfagp_upfirdn2d_bwd(const float* dy, const float* f, float* dx, int32_t N, int32_t C, int32_t H, int32_t W,
                        int32_t fh, int32_t fw, int32_t up, int32_t down, int32_t px0, int32_t px1, int32_t py0,
                        int32_t py1, float gain, void* stream) {
    HFAGP_REQUIRE(dy && f && dx, HFAGP_EBADARG, "upfirdn2d_bwd: null pointer");
    HFAGP_REQUIRE(up >= 1 && down >= 1 && fh >= 1 && fw >= 1, HFAGP_EBADARG, "upfirdn2d_bwd: bad up/down/filter");
    const int Ho = (H * up + py0 + py1 - fh) / down + 1, Wo = (W * up + px0 + px1 - fw) / down + 1;
    HFAGP_REQUIRE(Ho > 0 && Wo > 0, HFAGP_EBADARG, "upfirdn2d_bwd: empty forward output");
    // EG3D's Upfirdn2dCuda.backward: the same operator with up and down exchanged, the filter NOT flipped and the padding
    //   [fw - px0 - 1, W up - Wo down + px0 - up + 1, fh - py0 - 1, H up - Ho down + py0 - up + 1]; its output is [H][W]
    const long long total = (long long)N * C * H * W;
    upfirdn2d_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        dy, f, dx, N * C, Ho, Wo, fh, fw, down, up, fw - px0 - 1, fh - py0 - 1, H, W, gain, 1);
    return check_launch("upfirdn2d_bwd");
}

int hfagp_bias_act_fwd(const float* x, const float* b, float* y, int64_t n, int32_t C, int64_t inner, int32_t act,
                       float alpha, float gain, float clamp, void* stream) {
    HFAGP_REQUIRE(act == HFAGP_ACT_LINEAR || act == HFAGP_ACT_LRELU, HFAGP_EUNSUPPORTED, "bias_act: act %d", act);
    if (n == 0) return HFAGP_OK;                       // empty input: nothing to do (pointers may be null)
    HFAGP_REQUIRE(x && y && n > 0, HFAGP_EBADARG, "bias_act: null pointer");
    const long long blocks = (n + 255) / 256;
    bias_act_kernel<<<(unsigned)(blocks > 8192 ? 8192 : blocks), 256, 0, (hipStream_t)stream>>>(
        x, b, y, n, C > 0 ? C : 1, inner > 0 ? inner : 1, act, alpha, gain, clamp);
    return check_launch("bias_act");
}

int hfagp_bias_act_bwd(const float* dy, const float* y, float* dx, int64_t n, int32_t act, float alpha, float gain,
                       float clamp, void* stream) {
    HFAGP_REQUIRE(act == HFAGP_ACT_LINEAR || act == HFAGP_ACT_LRELU, HFAGP_EUNSUPPORTED, "bias_act_bwd: act %d", act);
    if (n == 0) return HFAGP_OK;
    HFAGP_REQUIRE(dy && y && dx && n > 0, HFAGP_EBADARG, "bias_act_bwd: null pointer");
    const long long blocks = (n + 255) / 256;
    bias_act_bwd_kernel<<<(unsigned)(blocks > 8192 ? 8192 : blocks), 256, 0, (hipStream_t)stream>>>(dy, y, dx, n, act, alpha,
                                                                                                 gain, clamp);
    return check_launch("bias_act_bwd");
}

int hfagp_nchw_to_nhwc(const float* x, float* y, int32_t B, int32_t C, int32_t H, int32_t W, void* stream) {
    HFAGP_REQUIRE(x && y, HFAGP_EBADARG, "nchw_to_nhwc: null pointer");
    dim3 grid((H * W + 31) / 32, (C + 31) / 32, B);
    transpose_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, y, C, H * W);
    return check_launch("nchw_to_nhwc");
}

int hfagp_nhwc_to_nchw(const float* x, float* y, int32_t B, int32_t C, int32_t H, int32_t W, void* stream) {
    HFAGP_REQUIRE(x && y, HFAGP_EBADARG, "nhwc_to_nchw: null pointer");
    dim3 grid((C + 31) / 32, (H * W + 31) / 32, B);
    transpose_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, y, H * W, C);
    return check_launch("nhwc_to_nchw");
}

}  // extern "C"

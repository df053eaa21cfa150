// Streaming kernels of the backward pass of the synthesis path (gfx950).  HBM-bound, channels-last,
// 16 bytes per lane.  The GEMM-shaped parts of the backward pass (conv bwd-data) reuse modconv_kernel
// with transposed weights (modes HFAGP_CONV3X3_BWD / HFAGP_CONVS2_BWD / HFAGP_CONV1X1).
//
// Gradient bookkeeping of one SynthesisLayer P (out = clamp(lrelu(d*conv(x*s, W) + noise + bias) * gain)):
//   g_pre  = g_out * gain * lrelu'(out) * [|out| < clamp]          (EG3D bias_act backward uses `out`)
//   g_conv = g_pre * d            -> input of P's bwd-data GEMM (adjoint of conv w.r.t. x*s)
//   dd[b,o] = sum_pix g_pre * conv,   conv = (pre - bias - noise) / d,  pre = lrelu^-1(out / gain)
//   dxs    = bwd-data(g_conv)     -> dx = dxs * s,   ds[b,i] = sum_pix dxs * x
// One fused pass per activation tensor X does the "dx of the consumers" and the "g_conv of the producer".
#include <algorithm>
#include "common.h"

namespace hfagp {

// reductions per channel: 0 ds_conv, 1 ds_rgb, 2 ds_small, 3 dd_p, 4 dbias_p = sum g_pre, 5 sum g_pre*noise,
// 6..9 sum_pix g_rgb_small[c] * x   (weight gradient rows of the small toRGB, before the style factor)
constexpr int kRed = 10;

// SMALL: the small (1 .. 4-channel) toRGB's adjoint is part of the pass; PG: parameter-gradient reductions (rows 4 .. 9).
// Round 5: compile-time variants — the generic kernel carried all ten reduction rows (40 registers) and the small-toRGB operands
// through every launch: 181 registers, two waves per SIMD; the common (frozen generator, 96-channel toRGB) instance needs three rows.
template <bool SMALL, bool PG, bool PACKED = false>
__global__ void __launch_bounds__(256, (SMALL && PACKED && !PG) ? 4 : 1) pointwise_bwd_kernel(const HfagpPointwiseBwdArgs a, int rows_per_block) {
    // block = (chunk of pixel rows, sample b); thread = (pixel lane, 4-channel group)
    // rows the variant really accumulates: 0 .. 3 without the parameter gradients (the LDS scratch of all ten rows was 40 KB per
    // workgroup: four workgroups per CU whatever the register count)
    constexpr int NR = PG ? kRed : 4;
    extern __shared__ __attribute__((aligned(16))) float red[];      // [256/C4][NR][C]
    const int C4 = a.C >> 2;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int c4 = threadIdx.x % C4, pl = threadIdx.x / C4, npl = 256 / C4;
    const int HW = a.H * a.W;
    const int p_begin = chunk * rows_per_block, p_end = min(HW, p_begin + rows_per_block);
    const size_t base = (size_t)b * HW * C4;

    float4 s_c = make_float4(0, 0, 0, 0), s_r = s_c, d_p = make_float4(1, 1, 1, 1), b_p = s_c;
    if (a.s_conv) s_c = reinterpret_cast<const float4*>(a.s_conv + (size_t)b * a.C)[c4];
    if (a.s_rgb) s_r = reinterpret_cast<const float4*>(a.s_rgb + (size_t)b * a.C)[c4];
    if (a.dcoef_p) d_p = reinterpret_cast<const float4*>(a.dcoef_p + (size_t)b * a.C)[c4];
    if (a.bias_p) b_p = reinterpret_cast<const float4*>(a.bias_p)[c4];
    float4 wsm[4];                       // small toRGB: w[c][i] * s_small[b][i]
    float4 s_sm = make_float4(0, 0, 0, 0);
    if constexpr (SMALL) {
        s_sm = reinterpret_cast<const float4*>(a.s_small + (size_t)b * a.C)[c4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            wsm[c] = c < a.Co ? reinterpret_cast<const float4*>(a.w_rgb_small + (size_t)c * a.C)[c4] : make_float4(0, 0, 0, 0);
    }
    float acc[kRed][4];
#pragma unroll
    for (int r = 0; r < kRed; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[r][k] = 0.f;

    const bool pvalid = pl < npl;        // 256 % C4 != 0 leaves idle threads
    const float nstr = a.noise_strength_dev ? *a.noise_strength_dev : a.noise_strength_p;      // (tuned generator: device value)
    // reciprocals once per thread instead of three IEEE divisions per element (~40 instructions each 4-channel group)
    const float rgain = 1.f / a.gain, ralpha = 1.f / a.alpha;
    const float dv[4] = {d_p.x, d_p.y, d_p.z, d_p.w}, bv[4] = {b_p.x, b_p.y, b_p.z, b_p.w};
    const float rdv[4] = {1.f / d_p.x, 1.f / d_p.y, 1.f / d_p.z, 1.f / d_p.w};
    // one pixel of this thread's 4 channels; the loads of TWO pixels are issued before either is used (round 4: one pixel
    // per iteration left one 16-byte load per operand in flight per thread — 2.0 - 2.4 TB/s on the 512^2 layers)
    auto pixel = [&](int p, const float4 x, const float4 gc, const float4 gr, const float4 gd, const float nraw,
                     const float gs0, const float gs1, const float gs2, const float gs3) __attribute__((always_inline)) {
        const size_t e = base + (size_t)p * C4 + c4;
        float gx[4] = {0, 0, 0, 0};
        const float xv[4] = {x.x, x.y, x.z, x.w};
        if (a.dxs_conv) {
            const float gv[4] = {gc.x, gc.y, gc.z, gc.w}, sv[4] = {s_c.x, s_c.y, s_c.z, s_c.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { gx[k] += gv[k] * sv[k]; acc[0][k] += gv[k] * xv[k]; }
        }
        if (!SMALL && a.dxs_rgb) {
            const float gv[4] = {gr.x, gr.y, gr.z, gr.w}, sv[4] = {s_r.x, s_r.y, s_r.z, s_r.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { gx[k] += gv[k] * sv[k]; acc[1][k] += gv[k] * xv[k]; }
        }
        if constexpr (SMALL) {
            float t[4] = {0, 0, 0, 0};
            const float gsm[4] = {gs0, gs1, gs2, gs3};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < a.Co) {
                    const float g = gsm[c];
                    t[0] += g * wsm[c].x; t[1] += g * wsm[c].y; t[2] += g * wsm[c].z; t[3] += g * wsm[c].w;
                    if constexpr (PG) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[6 + c][k] += g * xv[k];
                    }
                }
            const float sv[4] = {s_sm.x, s_sm.y, s_sm.z, s_sm.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) { gx[k] += t[k] * sv[k]; acc[2][k] += t[k] * xv[k]; }
        }
        if (a.g_direct) { gx[0] += gd.x; gx[1] += gd.y; gx[2] += gd.z; gx[3] += gd.w; }
        if (c4 == 0 && (a.g_nchw3_a || a.g_nchw3_b)) {       // image_raw = channels 0..2: NCHW gradients added in place
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const size_t q = ((size_t)b * 3 + k) * HW + p;
                if (a.g_nchw3_a) gx[k] += a.g_nchw3_a[q];
                if (a.g_nchw3_b) gx[k] += a.g_nchw3_b[q];
            }
        }
        // producer layer P: through clamp / gain / leaky-ReLU, then the demodulation
        const float nz = nraw * nstr;
        float go[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float g = gx[k];
            if (a.has_producer) {
                if (a.clamp >= 0.f && fabsf(xv[k]) >= a.clamp) g = 0.f;
                g *= a.gain;
                float pre = xv[k] * rgain;
                if (a.act_p == HFAGP_ACT_LRELU && !(xv[k] > 0.f)) { g *= a.alpha; pre *= ralpha; }   // (alpha at 0, as ATen / EG3D)
                acc[3][k] += g * (pre - bv[k] - nz) * rdv[k];
                if constexpr (PG) { acc[4][k] += g; acc[5][k] += g * nraw; }
                g *= dv[k];
            }
            go[k] = g;
        }
        reinterpret_cast<float4*>(a.g_out)[e] = make_float4(go[0], go[1], go[2], go[3]);
    };
    if (pvalid && c4 < C4) {
        const float4 z4 = make_float4(0, 0, 0, 0);
        const float4* X = reinterpret_cast<const float4*>(a.x);
        const float4* GC = reinterpret_cast<const float4*>(a.dxs_conv);
        const float4* GR = SMALL ? nullptr : reinterpret_cast<const float4*>(a.dxs_rgb);
        const float4* GD = reinterpret_cast<const float4*>(a.g_direct);
        // (the small toRGB's clamp mask [|y| < clamp] applied here when the caller hands over y: three framework kernels less)
        const bool mask_small = a.y_rgb_small != nullptr && a.clamp_rgb_small >= 0.f;
        auto small = [&](int p, int c) {
            if (!(SMALL && c < a.Co)) return 0.f;
            const size_t q = ((size_t)b * a.Co + c) * HW + p;
            const float g = a.g_rgb_small[q];
            return (mask_small && !(fabsf(a.y_rgb_small[q]) < a.clamp_rgb_small)) ? 0.f : g;
        };
        // the four (gradient, mask) pairs of a pixel are the same for all C / 4 lanes on it: with a power-of-two lane group lanes
        // 0 .. 3 of the group load the gradients, 4 .. 7 the masked outputs, and eight shuffles hand them round — instead of eight
        // scalar loads (and their 64-bit addresses: 18 registers) in every lane
        // (PACKED: the host checked that C / 4 is a power of two >= 8)
        constexpr bool packed = SMALL && PACKED;
        const int lane = threadIdx.x & 63, gbase = lane & ~(min(C4, 64) - 1), kq = lane & 7;
        auto small4 = [&](int p, float sv[4]) __attribute__((always_inline)) {
            if (!SMALL) { sv[0] = sv[1] = sv[2] = sv[3] = 0.f; return; }
            if constexpr (packed) {
                float v = 0.f;
                const int c = kq & 3;
                if (c < a.Co) {
                    const size_t q = ((size_t)b * a.Co + c) * HW + p;
                    if (kq < 4) v = a.g_rgb_small[q];
                    else if (mask_small) v = a.y_rgb_small[q];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float g = __shfl(v, gbase + k), y = __shfl(v, gbase + 4 + k);
                    sv[k] = (k < a.Co && !(mask_small && !(fabsf(y) < a.clamp_rgb_small))) ? g : 0.f;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) sv[k] = small(p, k);
            }
        };
        int p = p_begin + pl;
        for (; p + npl < p_end; p += 2 * npl) {
            const int q = p + npl;
            const size_t e0 = base + (size_t)p * C4 + c4, e1 = base + (size_t)q * C4 + c4;
            const float4 x0 = X[e0], x1 = X[e1];
            const float4 c0 = GC ? GC[e0] : z4, c1 = GC ? GC[e1] : z4;
            const float4 r0 = GR ? GR[e0] : z4, r1 = GR ? GR[e1] : z4;
            const float4 d0 = GD ? GD[e0] : z4, d1 = GD ? GD[e1] : z4;
            const float n0 = a.noise_p ? a.noise_p[p] : 0.f, n1 = a.noise_p ? a.noise_p[q] : 0.f;
            float s0[4], s1[4];
            small4(p, s0);
            small4(q, s1);
            pixel(p, x0, c0, r0, d0, n0, s0[0], s0[1], s0[2], s0[3]);
            pixel(q, x1, c1, r1, d1, n1, s1[0], s1[1], s1[2], s1[3]);
        }
        if (p < p_end) {
            const size_t e0 = base + (size_t)p * C4 + c4;
            float s0[4];
            small4(p, s0);
            pixel(p, X[e0], GC ? GC[e0] : z4, GR ? GR[e0] : z4, GD ? GD[e0] : z4, a.noise_p ? a.noise_p[p] : 0.f,
                  s0[0], s0[1], s0[2], s0[3]);
        }
    }
    // ---- block reduction over the pixel lanes, then one deterministic partial per (b, chunk)
    float* mine = red + ((size_t)pl * NR) * a.C + c4 * 4;
#pragma unroll
    for (int r = 0; r < NR; ++r)
        if (pvalid) *reinterpret_cast<float4*>(mine + r * a.C) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
    __syncthreads();
    for (int i = threadIdx.x; i < kRed * a.C; i += 256) {
        float s = 0.f;
        if (i < NR * a.C)
            for (int q = 0; q < npl; ++q) s += red[(size_t)q * NR * a.C + i];
        a.partial[(((size_t)b * gridDim.x + chunk) * kRed) * a.C + i] = s;
    }
}

// sums[b][r][c] = sum over chunks of partial[b][chunk][r][c].  Block = 16 consecutive outputs x 16 chunk lanes: lane q sums
// chunks q, q + 16, ... (eight loads in flight), then the 16 lane sums are added in a fixed tree — deterministic, and a
// thread's serial chain is nchunks / 16 loads (one thread per output walked all chunks: 13 us at 256 chunks, and the chunk
// count could not grow with the batch-1 layers that need more blocks).
__global__ void __launch_bounds__(256) reduce_partials_kernel(const float* __restrict__ partial,
                                                              float* __restrict__ sums, int B, int nchunks, int n) {
    __shared__ float red[16][17];
    const int kk = threadIdx.x & 15, q0 = threadIdx.x >> 4;
    const int k = blockIdx.x * 16 + kk, b = blockIdx.y;
    float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (k < n) {
        const float* src = partial + (size_t)b * nchunks * n + k;
        int q = q0;
        for (; q + 7 * 16 < nchunks; q += 8 * 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) part[u] += src[(size_t)(q + 16 * u) * n];
        }
        for (; q < nchunks; q += 16) part[0] += src[(size_t)q * n];
    }
    red[q0][kk] = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
    __syncthreads();
    if (q0 == 0 && k < n) {
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = red[u][kk];
#pragma unroll
        for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (u < w) t[u] += t[u + w];
        sums[(size_t)b * n + k] = t[0];
    }
}

// the same for the partial sums of SEVERAL fused passes in one launch (hfagp_reduce_partials_batch: a backward pass with the
// generator frozen needs its 19 sets of sums only at the very end, for the style gradients: 19 launches of 5 us -> 1)
constexpr int kRedBatchMax = 32;
struct RedBatch { HfagpReducePartialsItem it[kRedBatchMax]; };
__global__ void __launch_bounds__(256) reduce_partials_batch_kernel(const RedBatch t) {
    // block = 32 consecutive outputs x 8 chunk lanes (128-byte rows; eight loads in flight per lane)
    __shared__ float red[8][33];
    const HfagpReducePartialsItem& a = t.it[blockIdx.z];
    const int kk = threadIdx.x & 31, q0 = threadIdx.x >> 5;
    const int k = blockIdx.x * 32 + kk, b = blockIdx.y;
    if (b >= a.B || blockIdx.x * 32 >= a.n) return;              // (uniform per workgroup)
    // (32 loads in flight per lane: with 8 a lane's 80 chunks of the 512^2 passes were ten memory round trips — 50 us for 40 MB)
    float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (k < a.n) {
        const float* src = a.partial + (size_t)b * a.nchunks * a.n + k;
        for (int q = q0; q < a.nchunks; q += 8 * 32) {
            float v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = q + 8 * u < a.nchunks ? src[(size_t)(q + 8 * u) * a.n] : 0.f;
#pragma unroll
            for (int u = 0; u < 32; ++u) part[u & 7] += v[u];
        }
    }
    red[q0][kk] = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
    __syncthreads();
    if (q0 == 0 && k < a.n) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = red[u][kk];
        a.sums[(size_t)b * a.n + k] = ((v[0] + v[4]) + (v[2] + v[6])) + ((v[1] + v[5]) + (v[3] + v[7]));
    }
}

// depth[i] = clamp(depth[i], min_j t[j].x, max_j t[j].y) for the whole batch in one launch (hfagp_depth_clamp): every block
// reduces ALL min / max pairs itself (they stay in L2: 256 KB at two frames, 4 MB at 32) and clamps its own slice
__global__ void __launch_bounds__(1024) depth_clamp_kernel(float* __restrict__ depth, const float2* __restrict__ t, int n) {
    __shared__ float smin[16], smax[16];
    float lo = INFINITY, hi = -INFINITY;
    // two pairs per 16-byte load, eight loads in flight per thread (round 6: one 8-byte load per iteration made the 512 iterations
    // of a 32-frame batch a chain of L2 round trips: 202 us for 4 MB)
    const int head = (reinterpret_cast<uintptr_t>(t) & 8) ? 1 : 0;        // (an 8-byte aligned caller buffer: its first pair alone)
    if (head && threadIdx.x == 0) { lo = t[0].x; hi = t[0].y; }
    t += head; n -= head;
    const float4* t4 = reinterpret_cast<const float4*>(t);
    const int n4 = n >> 1;
    int i = threadIdx.x;
    for (; i + 7 * 1024 < n4; i += 8 * 1024) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = t4[i + u * 1024];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            lo = fminf(lo, fminf(v[u].x, v[u].z));
            hi = fmaxf(hi, fmaxf(v[u].y, v[u].w));
        }
    }
    for (; i < n4; i += 1024) {
        const float4 v = t4[i];
        lo = fminf(lo, fminf(v.x, v.z));
        hi = fmaxf(hi, fmaxf(v.y, v.w));
    }
    if ((n & 1) && threadIdx.x == 0) { lo = fminf(lo, t[n - 1].x); hi = fmaxf(hi, t[n - 1].y); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    lo = smin[0]; hi = smax[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) { lo = fminf(lo, smin[w]); hi = fmaxf(hi, smax[w]); }
    // no second launch, no grid barrier (one block alone took 21 us at two frames)
    n += head;
    for (int j = blockIdx.x * 1024 + threadIdx.x; j < n; j += gridDim.x * 1024) depth[j] = fminf(fmaxf(depth[j], lo), hi);
}

// ---------------------------------------------------------------- adjoint of (FIR pad 1 gain 4) + parity split
// g_y [B][2H][2W][C] -> gph [2][2][B][H+1][W+1][C],  gph[a][b][m][n] = g_yt[2m+a][2n+b],
// g_yt[Y][X] = sum_{p,q} f[p] f[q] g_y[Y-p+1][X-q+1],  f = [1,3,3,1]/4.
// Separable sliding window like upfir_epilogue_kernel: a thread owns 4 channels of two adjacent g_yt columns and
// walks a strip of rows; 5 input columns per input row feed both columns and every input row of the strip is
// read once (3.4 loads per output instead of 16); borders by clamped addresses and zeroed tap weights.
constexpr int kBwdStrip = 8;

__global__ void __launch_bounds__(256) upfir_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gph,
                                                        int B, int H, int W, int C) {
    const int C4 = C >> 2;
    const int Ho = 2 * H, Wo = 2 * W, Hi = 2 * H + 1;          // g_y is Ho x Wo, g_yt is Hi x (Wo + 1)
    const int Wp = W + 1;                                       // column pairs (X, X+1), X = 2n
    const int strips = (Hi + 1 + kBwdStrip - 1) / kBwdStrip;    // rows 0 .. Hi (row Hi = the zero padding of a = 1)
    const long long total = (long long)B * strips * Wp * C4;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= total) return;
    const int c4 = (int)(tid % C4);
    const int n = (int)((tid / C4) % Wp);
    const int st = (int)((tid / ((long long)C4 * Wp)) % strips);
    const int b = (int)(tid / ((long long)C4 * Wp * strips));
    const int X = 2 * n, Y0 = st * kBwdStrip;
    const float4* src = reinterpret_cast<const float4*>(gy) + (size_t)b * Ho * Wo * C4 + c4;
    const float f0 = 0.25f, f1 = 0.75f;

    // input columns X-2 .. X+2
    int xo[5]; float xm[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int x = X + q - 2;
        xm[q] = (x >= 0 && x < Wo) ? 1.f : 0.f;
        xo[q] = min(max(x, 0), Wo - 1) * C4;
    }
    // column X: inputs X-2, X-1, X, X+1 with f3, f2, f1, f0;  column X+1: inputs X-1 .. X+2
    const float wa[4] = {f0 * xm[0], f1 * xm[1], f1 * xm[2], f0 * xm[3]};
    const float wb[4] = {f0 * xm[1], f1 * xm[2], f1 * xm[3], f0 * xm[4]};
    const bool colb = X + 1 <= Wo;                              // column X+1 exists in g_yt (else: zero padding)

    auto hrow = [&](int y, float4& ha, float4& hb) {
        const float m = (y >= 0 && y < Ho) ? 1.f : 0.f;
        const float4* row = src + (size_t)min(max(y, 0), Ho - 1) * Wo * C4;
        float4 v[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) v[q] = row[xo[q]];
        ha.x = m * (wa[0] * v[0].x + wa[1] * v[1].x + wa[2] * v[2].x + wa[3] * v[3].x);
        ha.y = m * (wa[0] * v[0].y + wa[1] * v[1].y + wa[2] * v[2].y + wa[3] * v[3].y);
        ha.z = m * (wa[0] * v[0].z + wa[1] * v[1].z + wa[2] * v[2].z + wa[3] * v[3].z);
        ha.w = m * (wa[0] * v[0].w + wa[1] * v[1].w + wa[2] * v[2].w + wa[3] * v[3].w);
        hb.x = m * (wb[0] * v[1].x + wb[1] * v[2].x + wb[2] * v[3].x + wb[3] * v[4].x);
        hb.y = m * (wb[0] * v[1].y + wb[1] * v[2].y + wb[2] * v[3].y + wb[3] * v[4].y);
        hb.z = m * (wb[0] * v[1].z + wb[1] * v[2].z + wb[2] * v[3].z + wb[3] * v[4].z);
        hb.w = m * (wb[0] * v[1].w + wb[1] * v[2].w + wb[2] * v[3].w + wb[3] * v[4].w);
    };

    // g_yt[Y] = f0 h[Y+1] + f1 h[Y] + f1 h[Y-1] + f0 h[Y-2]
    float4 a0, a1, a2, b0, b1, b2;
    hrow(Y0 - 2, a0, b0); hrow(Y0 - 1, a1, b1); hrow(Y0, a2, b2);
    const long long img = (long long)B * (H + 1) * (W + 1) * C4;         // float4 per parity image
    float4* dst = reinterpret_cast<float4*>(gph) + ((size_t)b * (H + 1) * (W + 1) + n) * C4 + c4;
#pragma unroll
    for (int k = 0; k < kBwdStrip; ++k) {
        const int Y = Y0 + k;
        if (Y > Hi) break;                                      // rows 0 .. Hi inclusive are stored
        float4 a3, b3;
        hrow(Y + 1, a3, b3);
        float4 oa, ob;
        oa.x = f0 * a3.x + f1 * a2.x + f1 * a1.x + f0 * a0.x;
        oa.y = f0 * a3.y + f1 * a2.y + f1 * a1.y + f0 * a0.y;
        oa.z = f0 * a3.z + f1 * a2.z + f1 * a1.z + f0 * a0.z;
        oa.w = f0 * a3.w + f1 * a2.w + f1 * a1.w + f0 * a0.w;
        ob.x = f0 * b3.x + f1 * b2.x + f1 * b1.x + f0 * b0.x;
        ob.y = f0 * b3.y + f1 * b2.y + f1 * b1.y + f0 * b0.y;
        ob.z = f0 * b3.z + f1 * b2.z + f1 * b1.z + f0 * b0.z;
        ob.w = f0 * b3.w + f1 * b2.w + f1 * b1.w + f0 * b0.w;
        if (Y > Ho) oa = make_float4(0.f, 0.f, 0.f, 0.f);       // row Hi of the a = 1 images: zero padding
        if (Y > Ho || !colb) ob = make_float4(0.f, 0.f, 0.f, 0.f);
        const int pa = Y & 1, m = Y >> 1;                       // parity image (pa, 0 / 1), row m
        float4* row = dst + (size_t)(2 * pa) * img + (size_t)m * (W + 1) * C4;
        row[0] = oa;                                            // image (pa, 0), column n
        row[img] = ob;                                          // image (pa, 1), column n
        a0 = a1; a1 = a2; a2 = a3;
        b0 = b1; b1 = b2; b2 = b3;
    }
}

// ---------------------------------------------------------------- adjoint of upsample2d (skip images)
// g_in[i][j] = sum_{p,q} k[p] k[q] g[2i-1+p][2j-1+q],  k = [.25,.75,.75,.25]; generic strides so that the same
// kernel serves channels-last (inner = C) and NCHW (inner = 1) tensors.
__global__ void __launch_bounds__(256) upsample2d_bwd_kernel(const float* __restrict__ g, float* __restrict__ gin,
                                                             long long outer, int H, int W, int inner) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= outer * H * W * inner) return;
    const int c = (int)(tid % inner);
    const int j = (int)((tid / inner) % W);
    const int i = (int)((tid / ((long long)inner * W)) % H);
    const long long o = tid / ((long long)inner * W * H);
    const int Ho = 2 * H, Wo = 2 * W;
    const float k[4] = {0.25f, 0.75f, 0.75f, 0.25f};
    const float* src = g + (size_t)o * Ho * Wo * inner + c;
    float acc = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int y = 2 * i - 1 + p;
        if (y < 0 || y >= Ho) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int x = 2 * j - 1 + q;
            if (x < 0 || x >= Wo) continue;
            acc += k[p] * k[q] * src[((size_t)y * Wo + x) * inner];
        }
    }
    gin[tid] = acc;
}

// channels-last with inner % 4 == 0 (the 96-channel skip images of the backbone): 16 bytes per lane, 32-bit index arithmetic
// (the scalar form above ran the 256^2 x 96 image at 1.2 TB/s: 16 dword loads and three 64-bit divisions per output float)
__global__ void __launch_bounds__(256) upsample2d_bwd_v4_kernel(const float4* __restrict__ g, float4* __restrict__ gin,
                                                                int outer, int H, int W, int inner4) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= outer * H * W * inner4) return;
    const int c = tid % inner4, j = (tid / inner4) % W, i = (tid / (inner4 * W)) % H, o = tid / (inner4 * W * H);
    const int Ho = 2 * H, Wo = 2 * W;
    const float k[4] = {0.25f, 0.75f, 0.75f, 0.25f};
    const float4* src = g + (size_t)o * Ho * Wo * inner4 + c;
    float4 v[4][4];
    float wy[4], wx[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int y = 2 * i - 1 + p, x = 2 * j - 1 + p;
        wy[p] = (y >= 0 && y < Ho) ? k[p] : 0.f;
        wx[p] = (x >= 0 && x < Wo) ? k[p] : 0.f;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int y = min(max(2 * i - 1 + p, 0), Ho - 1), x = min(max(2 * j - 1 + q, 0), Wo - 1);      // (clamped: the weight is 0)
            v[p][q] = src[((size_t)y * Wo + x) * inner4];
        }
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float w = wy[p] * wx[q];
            acc.x += w * v[p][q].x; acc.y += w * v[p][q].y; acc.z += w * v[p][q].z; acc.w += w * v[p][q].w;
        }
    gin[tid] = acc;
}

// plane-major [B][3][H][W][Cp] -> channels-last [B][H][W][3*Cp] (gradient of the tri-plane volume back
// into the layout of the backbone's skip image)
__global__ void __launch_bounds__(256) planes_to_nhwc_kernel(const float* __restrict__ pm, float* __restrict__ y,
                                                             int B, int H, int W, int Cp) {
    const int C4 = (3 * Cp) >> 2;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long long)B * H * W * C4) return;
    const int c4 = (int)(tid % C4);
    const long long pix = tid / C4;
    const int x = (int)(pix % W), yy = (int)((pix / W) % H), b = (int)(pix / ((long long)W * H));
    const int c = c4 * 4, pl = c / Cp, ch = c % Cp;
    reinterpret_cast<float4*>(y)[tid] =
        *reinterpret_cast<const float4*>(pm + ((((size_t)b * 3 + pl) * H + yy) * W + x) * Cp + ch);
}

// ---------------------------------------------------------------- styles -> latent
// dstot[b][i] = gain * ( ds[b][i] - s[b][i]/gain... ) see host comment; wave per (b, i)
__global__ void __launch_bounds__(256) style_bwd_ds_kernel(const float* __restrict__ ds, const float* __restrict__ dd,
                                                           const float* __restrict__ styles,
                                                           const float* __restrict__ dcoef,
                                                           const float* __restrict__ wsq, float* __restrict__ dstot,
                                                           int B, int Cin, int Cout, float style_gain) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= B * Cin) return;
    const int b = wave / Cin, i = wave % Cin;
    float acc = 0.f;
    if (dd)   // d = rsqrt(sum_i s_i^2 wsq[o][i] + eps)  ->  dd/ds_i = -d^3 s_i wsq[o][i]
        for (int o = lane; o < Cout; o += 64) {
            const float d = dcoef[(size_t)b * Cout + o];
            acc += dd[(size_t)b * Cout + o] * d * d * d * wsq[(size_t)o * Cin + i];
        }
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) dstot[wave] = (ds[wave] - styles[wave] * acc) * style_gain;
}

// dw[b][k] (+)= wgain * sum_i dstot[b][i] * A[i][k].  Block = 64 columns k x 4 row groups; coalesced rows of A.
__global__ void __launch_bounds__(256) style_bwd_dw_kernel(const float* __restrict__ dstot, const float* __restrict__ A,
                                                           float* __restrict__ dw, int B, int Cin, int w_dim,
                                                           int dw_stride, float wgain, int accumulate) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, k = blockIdx.x * 64 + (threadIdx.x & 63), ig = threadIdx.x >> 6;
    float acc = 0.f;
    if (k < w_dim) {
        float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // 8 independent load chains in flight
        int i = ig;
        for (; i + 28 < Cin; i += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                part[u] += dstot[(size_t)b * Cin + i + 4 * u] * A[(size_t)(i + 4 * u) * w_dim + k];
        }
        for (; i < Cin; i += 4) part[0] += dstot[(size_t)b * Cin + i] * A[(size_t)i * w_dim + k];
        acc = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
    }
    red[ig][threadIdx.x & 63] = acc;
    __syncthreads();
    if (ig == 0 && k < w_dim) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) * wgain;
        float* dst = dw + (size_t)b * dw_stride + k;
        *dst = accumulate ? *dst + v : v;
    }
}

// ---------------------------------------------------------------- all style gradients of a backward pass, batched
// The per-layer calls (52 launches of ~5-9 us per fitting step, each fed by tiny `.contiguous()` copies of strided
// views) become two launches: item = one affine layer; ds / dd are read with a row stride; the d_ws rows are
// accumulated by ONE block per (row, sample, 64 columns) that walks the items of its row in order (several layers
// share a ws row: deterministic, no race).
constexpr int kStyleBwdBatch = 32;
struct StyleBwdBatch {
    HfagpStyleBwdItem it[kStyleBwdBatch];
    int row_start[kStyleBwdBatch + 1];     // items are sorted by dw pointer: row r owns items [row_start[r], row_start[r+1])
};

__global__ void __launch_bounds__(256) style_bwd_ds_batch_kernel(const StyleBwdBatch t) {
    const HfagpStyleBwdItem& a = t.it[blockIdx.y];
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= a.B * a.Cin) return;
    const int b = wave / a.Cin, i = wave % a.Cin;
    float acc = 0.f;
    if (a.dd)
        for (int o = lane; o < a.Cout; o += 64) {
            const float d = a.dcoef[(size_t)b * a.Cout + o];
            acc += a.dd[(size_t)b * a.dd_stride + o] * d * d * d * a.wsq[(size_t)o * a.Cin + i];
        }
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) a.dstot[wave] = (a.ds[(size_t)b * a.ds_stride + i] - a.styles[wave] * acc) * a.style_gain;
}

__global__ void __launch_bounds__(256) style_bwd_dw_batch_kernel(const StyleBwdBatch t) {
    __shared__ float red[4][64];
    const int row = blockIdx.z, b = blockIdx.y, k = blockIdx.x * 64 + (threadIdx.x & 63), ig = threadIdx.x >> 6;
    const int i0 = t.row_start[row], i1 = t.row_start[row + 1];
    const HfagpStyleBwdItem& first = t.it[i0];
    float acc = 0.f;
    if (k < first.w_dim)
        for (int it = i0; it < i1; ++it) {
            const HfagpStyleBwdItem& a = t.it[it];
            float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int i = ig;
            for (; i + 28 < a.Cin; i += 32) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    part[u] += a.dstot[(size_t)b * a.Cin + i + 4 * u] * a.affine_w[(size_t)(i + 4 * u) * a.w_dim + k];
            }
            for (; i < a.Cin; i += 4) part[0] += a.dstot[(size_t)b * a.Cin + i] * a.affine_w[(size_t)i * a.w_dim + k];
            acc += ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
        }
    red[ig][threadIdx.x & 63] = acc;
    __syncthreads();
    if (ig == 0 && k < first.w_dim) {
        const float v = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) *
                        rsqrtf((float)first.w_dim);
        float* dst = first.dw + (size_t)b * first.dw_stride + k;
        *dst += v;
    }
}

}  // namespace hfagp

using namespace hfagp;

extern "C" {

int hfagp_pointwise_bwd(const HfagpPointwiseBwdArgs* a, void* stream) {
    HFAGP_REQUIRE(a && a->x && a->g_out && a->partial, HFAGP_EBADARG, "pointwise_bwd: null pointer");
    HFAGP_REQUIRE(a->C % 4 == 0 && a->C / 4 <= 256 && a->B > 0 && a->H > 0 && a->W > 0, HFAGP_EUNSUPPORTED,
                  "pointwise_bwd: C=%d must be a multiple of 4 and <= 1024", a->C);
    HFAGP_REQUIRE(a->noise_strength_p == a->noise_strength_p, HFAGP_EBADARG, "pointwise_bwd: NaN noise strength");
    HFAGP_REQUIRE(!a->g_rgb_small || (a->Co >= 1 && a->Co <= 4 && a->w_rgb_small && a->s_small), HFAGP_EBADARG,
                  "pointwise_bwd: small toRGB needs 1..4 channels, weights and styles");
    HFAGP_REQUIRE(a->nchunks >= 1, HFAGP_EBADARG, "pointwise_bwd: nchunks");
    // (a layer has ONE toRGB: the 96-channel one arrives as dxs_rgb, the 1 .. 4-channel one as g_rgb_small; the small variants are
    // compiled without the dxs_rgb operand: 18 registers, three resident workgroups per CU instead of two with parameter gradients)
    HFAGP_REQUIRE(!(a->g_rgb_small && a->dxs_rgb), HFAGP_EUNSUPPORTED, "pointwise_bwd: dxs_rgb and g_rgb_small are exclusive");
    const int HW = a->H * a->W;
    const int rows = (HW + a->nchunks - 1) / a->nchunks;
    const int npl = 256 / (a->C / 4);
    const size_t lds = (size_t)npl * (a->param_grads ? kRed : 4) * a->C * sizeof(float);
    HFAGP_REQUIRE(lds <= 64 * 1024, HFAGP_EUNSUPPORTED, "pointwise_bwd: LDS %zu", lds);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(a->nchunks, a->B);
    if (a->g_rgb_small) {
        const int c4 = a->C / 4;
        const bool packed = c4 >= 8 && (c4 & (c4 - 1)) == 0;
        if (a->param_grads) {
            if (packed) pointwise_bwd_kernel<true, true, true><<<grid, 256, lds, s>>>(*a, rows);
            else pointwise_bwd_kernel<true, true><<<grid, 256, lds, s>>>(*a, rows);
        } else {
            if (packed) pointwise_bwd_kernel<true, false, true><<<grid, 256, lds, s>>>(*a, rows);
            else pointwise_bwd_kernel<true, false><<<grid, 256, lds, s>>>(*a, rows);
        }
    } else {
        if (a->param_grads) pointwise_bwd_kernel<false, true><<<grid, 256, lds, s>>>(*a, rows);
        else pointwise_bwd_kernel<false, false><<<grid, 256, lds, s>>>(*a, rows);
    }
    const int n = kRed * a->C;
    // (sums NULL, ABI 12: the caller reduces `partial` later, with other passes' in one launch: hfagp_reduce_partials_batch)
    if (a->sums) reduce_partials_kernel<<<dim3((n + 15) / 16, a->B), 256, 0, s>>>(a->partial, a->sums, a->B, a->nchunks, n);
    return check_launch("pointwise_bwd");
}

int hfagp_reduce_partials_batch(const HfagpReducePartialsItem* items, int32_t n, void* stream) {
    HFAGP_REQUIRE(items && n >= 1 && n <= kRedBatchMax, HFAGP_EBADARG, "reduce_partials_batch: 1..%d items", kRedBatchMax);
    RedBatch t;
    int most_n = 0, most_b = 0;
    for (int i = 0; i < n; ++i) {
        const HfagpReducePartialsItem& a = items[i];
        HFAGP_REQUIRE(a.partial && a.sums && a.B > 0 && a.nchunks > 0 && a.n > 0, HFAGP_EBADARG, "reduce_partials_batch: item %d", i);
        t.it[i] = a;
        most_n = std::max(most_n, a.n);
        most_b = std::max(most_b, a.B);
    }
    reduce_partials_batch_kernel<<<dim3((unsigned)((most_n + 31) / 32), (unsigned)most_b, (unsigned)n), 256, 0, (hipStream_t)stream>>>(t);
    return check_launch("reduce_partials_batch");
}

int hfagp_depth_clamp(float* depth, const float* tminmax, int64_t n, void* stream) {
    HFAGP_REQUIRE(depth && tminmax && n > 0, HFAGP_EBADARG, "depth_clamp: null pointer / n <= 0");
    HFAGP_REQUIRE(n <= (1ll << 30), HFAGP_EUNSUPPORTED, "depth_clamp: n=%lld rays", (long long)n);
    // (round 5: any batch — the cap of 65 536 rays left ten framework launches behind the ray march at B = 32; up to 64 blocks,
    // each re-reading the n pairs from L2: 64 x 4 MB at 32 frames)
    const unsigned blocks = (unsigned)std::min<long long>(64, (n + 4095) / 4096);
    depth_clamp_kernel<<<blocks, 1024, 0, (hipStream_t)stream>>>(depth, reinterpret_cast<const float2*>(tminmax), (int)n);
    return check_launch("depth_clamp");
}

int hfagp_upfir_bwd(const float* gy, float* gph, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
    HFAGP_REQUIRE(gy && gph, HFAGP_EBADARG, "upfir_bwd: null pointer");
    HFAGP_REQUIRE(C % 4 == 0 && B > 0 && H > 0 && W > 0, HFAGP_EUNSUPPORTED, "upfir_bwd: C=%d", C);
    const int strips = (2 * H + 2 + kBwdStrip - 1) / kBwdStrip;
    const long long total = (long long)B * strips * (W + 1) * (C / 4);
    upfir_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(gy, gph, B, H, W, C);
    return check_launch("upfir_bwd");
}

int hfagp_upsample2d_bwd(const float* g, float* gin, int64_t outer, int32_t H, int32_t W, int32_t inner, void* stream) {
    HFAGP_REQUIRE(g && gin, HFAGP_EBADARG, "upsample2d_bwd: null pointer");
    HFAGP_REQUIRE(outer > 0 && H > 0 && W > 0 && inner > 0, HFAGP_EBADARG, "upsample2d_bwd: bad dims");
    const long long total = outer * H * W * inner;
    if (inner % 4 == 0 && total * 4 < (1ll << 31)) {
        upsample2d_bwd_v4_kernel<<<(unsigned)((total / 4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(
            reinterpret_cast<const float4*>(g), reinterpret_cast<float4*>(gin), (int)outer, H, W, inner / 4);
        return check_launch("upsample2d_bwd");
    }
    upsample2d_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(g, gin, outer, H, W, inner);
    return check_launch("upsample2d_bwd");
}

int hfagp_planes_to_nhwc(const float* pm, float* y, int32_t B, int32_t H, int32_t W, int32_t Cp, void* stream) {
    HFAGP_REQUIRE(pm && y, HFAGP_EBADARG, "planes_to_nhwc: null pointer");
    HFAGP_REQUIRE(Cp % 4 == 0, HFAGP_EUNSUPPORTED, "planes_to_nhwc: Cp=%d", Cp);
    const long long total = (long long)B * H * W * (3 * Cp / 4);
    planes_to_nhwc_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(pm, y, B, H, W, Cp);
    return check_launch("planes_to_nhwc");
}

int hfagp_style_batch_bwd(const HfagpStyleBwdItem* items, int32_t n, void* stream) {
    HFAGP_REQUIRE(items && n >= 1 && n <= kStyleBwdBatch, HFAGP_EBADARG, "style_batch_bwd: 1..%d items", kStyleBwdBatch);
    StyleBwdBatch t;
    int max_waves = 0, nrows = 0;
    for (int i = 0; i < n; ++i) {
        const HfagpStyleBwdItem& a = items[i];
        HFAGP_REQUIRE(a.ds && a.styles && a.affine_w && a.dstot && a.dw, HFAGP_EBADARG, "style_batch_bwd: null pointer (item %d)", i);
        HFAGP_REQUIRE(!a.dd || (a.dcoef && a.wsq && a.Cout > 0), HFAGP_EBADARG, "style_batch_bwd: dd needs dcoef and wsq (item %d)", i);
        HFAGP_REQUIRE(a.B == items[0].B && a.w_dim == items[0].w_dim, HFAGP_EBADARG, "style_batch_bwd: B / w_dim differ (item %d)", i);
        HFAGP_REQUIRE(i == 0 || a.dw >= items[i - 1].dw, HFAGP_EBADARG, "style_batch_bwd: items must be sorted by dw (item %d)", i);
        t.it[i] = a;
        if (i == 0 || a.dw != items[i - 1].dw) t.row_start[nrows++] = i;
        max_waves = a.B * a.Cin > max_waves ? a.B * a.Cin : max_waves;
    }
    t.row_start[nrows] = n;
    hipStream_t s = (hipStream_t)stream;
    style_bwd_ds_batch_kernel<<<dim3((max_waves + 3) / 4, n), 256, 0, s>>>(t);
    style_bwd_dw_batch_kernel<<<dim3((items[0].w_dim + 63) / 64, items[0].B, nrows), 256, 0, s>>>(t);
    return check_launch("style_batch_bwd");
}

int hfagp_style_bwd(const HfagpStyleBwdArgs* a, void* stream) {
    HFAGP_REQUIRE(a && a->ds && a->styles && a->affine_w && a->dstot && a->dw, HFAGP_EBADARG, "style_bwd: null pointer");
    HFAGP_REQUIRE(!a->dd || (a->dcoef && a->wsq && a->Cout > 0), HFAGP_EBADARG, "style_bwd: dd needs dcoef and wsq");
    hipStream_t s = (hipStream_t)stream;
    const int waves = a->B * a->Cin;
    style_bwd_ds_kernel<<<(waves + 3) / 4, 256, 0, s>>>(a->ds, a->dd, a->styles, a->dcoef, a->wsq, a->dstot, a->B,
                                                       a->Cin, a->Cout, a->style_gain);
    style_bwd_dw_kernel<<<dim3((a->w_dim + 63) / 64, a->B), 256, 0, s>>>(a->dstot, a->affine_w, a->dw, a->B, a->Cin, a->w_dim,
                                                       a->dw_stride, 1.0f / sqrtf((float)a->w_dim), a->accumulate);
    return check_launch("style_bwd");
}

}  // extern "C"

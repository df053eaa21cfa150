// Weight gradient of the modulated convolutions (gfx950, v_mfma_f32_32x32x2_f32, exact fp32).
//
//   dW[t][ci][co] = sum_{b, m, n}  (x * s_b)[b][m + ady_t][n + adx_t][ci] * g[img_t][b][m + bdy_t][n + bdx_t][co]
//
// GEMM view per tap: M = ci, N = co, K = B*H*W positions.  A block owns a 64(ci) x 64(co) tile for ALL taps
// (each wave a 32x32 tile x ntaps accumulators) and walks 4x16-position tiles: the x patch and the g patch
// (both with a 1-pixel halo) are staged in LDS once per position tile and reused by the 9 taps.  Split-K over
// (b, position tile) writes deterministic slabs; wgrad_reduce_kernel sums them, adds the demodulation term
// and writes the gradient in the parameter's [Cout][Cin][kh][kw] layout.
//   3x3 conv     : x shifts with the tap (ady, adx in -1..1), g does not.
//   up-conv      : g = the four parity images of the y_t gradient (hfagp_upfir_bwd); tap (ti,tj) reads parity
//                  (ti&1, tj&1) at shift (ti>>1, tj>>1); x does not shift.
//   1x1 (toRGB)  : one tap, no shifts.
#include "common.h"

namespace hfagp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TPH = 2, TPW = 16;                 // position tile (2 x 16: 78 KB of LDS double-buffered -> 2 blocks per CU)
constexpr int PPH = TPH + 2, PPW = TPW + 2;      // patch with halo
constexpr int WT = 64;                           // ci / co tile
constexpr int WS = WT + 4;                       // LDS row stride (floats)

struct WTap { signed char ady, adx, bdy, bdx, widx; };

struct WgradParams {
    const float* x; const float* styles; const float* g;
    float* slabs;                                // [nunits_split][ntaps][Cin][Cout]
    int B, H, W, Cin, Cout;                      // x is [B][H][W][Cin]
    int gH, gW;                                  // g images are [B][gH][gW][Cout]
    int ntaps, tiles_h, tiles_w, ksplit;
    WTap tap[9];
};

// Double-buffered: the x / g patches of position tile u+1 are fetched into registers before the MFMAs of tile u
// and written to the other LDS buffer after them, so there is ONE barrier per tile and the global latency hides
// under 288 MFMAs.  117 KB of LDS -> one workgroup (one wave per SIMD) per CU, which is enough to keep the
// matrix pipe busy because the 9 tap accumulators are independent.
constexpr int STG = (PPH * PPW * 16 + 255) / 256;      // float4 per thread per patch

// SHARE = 1: all taps read the same g element (3x3 / 1x1: only x shifts) -> one B fetch per K step;
// SHARE = 2: all taps read the same x element (up-conv parity launches: only g shifts) -> one A fetch per K step.
// (halves the ds_read traffic per MFMA, which is what bounds this kernel: 2 dwords per lane per MFMA otherwise)
template <int NT, int SHARE>
__global__ void __launch_bounds__(256, 2) wgrad_kernel(const WgradParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PATCH = PPH * PPW * WS;
    // buffer k: As = lds + k*2*PATCH, Bs = As + PATCH
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;     // wave tile: ci rows 32*wi.., co cols 32*wj..
    const int h = lane >> 5, l31 = lane & 31;
    const int ci0 = blockIdx.x * WT, co0 = blockIdx.y * WT, ks = blockIdx.z;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int units = p.B * p.tiles_h * p.tiles_w;
    const int u_begin = (int)(((long long)units * ks) / p.ksplit), u_end = (int)(((long long)units * (ks + 1)) / p.ksplit);
    // Staging loads are unconditional (an out-of-range slot reads element 0 and is multiplied by a 0 mask when it is
    // committed) and nothing in fetch() consumes a loaded value, so the loads of tile u+1 stay in flight under the MFMAs
    // of tile u; the style is applied at commit time.  (Not for the nine-accumulator variant: the extra staging
    // registers made it spill — it applies style and mask in fetch() as before.)
    constexpr bool DEFER = NT != 9;
    float4 ra[STG], rb[STG], rs[DEFER ? STG : 1];
    float ma[DEFER ? STG : 1], mb[DEFER ? STG : 1];

    auto fetch = [&](int u) {
        const int tw = u % p.tiles_w, th = (u / p.tiles_w) % p.tiles_h, b = u / (p.tiles_w * p.tiles_h);
        const int m0 = th * TPH, n0 = tw * TPW;
#pragma unroll
        for (int k = 0; k < STG; ++k) {
            const int idx = min(tid + k * 256, PPH * PPW * 16 - 1);
            const int pix = idx >> 4, q = idx & 15;
            const int iy = m0 - 1 + pix / PPW, ix = n0 - 1 + pix % PPW;
            const bool oka = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W && ci0 + 4 * q < p.Cin;
            const bool okb = iy >= 0 && iy < p.gH && ix >= 0 && ix < p.gW && co0 + 4 * q < p.Cout;
            const float4 va = *reinterpret_cast<const float4*>(p.x + (oka ? (((size_t)b * p.H + iy) * p.W + ix) * p.Cin + ci0 + 4 * q : 0));
            const float4 vb = *reinterpret_cast<const float4*>(p.g + (okb ? (((size_t)b * p.gH + iy) * p.gW + ix) * p.Cout + co0 + 4 * q : 0));
            const float4 sv = (p.styles && oka) ? *reinterpret_cast<const float4*>(p.styles + (size_t)b * p.Cin + ci0 + 4 * q)
                                                : make_float4(1.f, 1.f, 1.f, 1.f);
            if constexpr (DEFER) {
                ma[k] = oka ? 1.f : 0.f;
                mb[k] = okb ? 1.f : 0.f;
                ra[k] = va; rb[k] = vb; rs[k] = sv;
            } else {
                const float fa = oka ? 1.f : 0.f, fb = okb ? 1.f : 0.f;
                ra[k] = make_float4(va.x * (sv.x * fa), va.y * (sv.y * fa), va.z * (sv.z * fa), va.w * (sv.w * fa));
                rb[k] = make_float4(vb.x * fb, vb.y * fb, vb.z * fb, vb.w * fb);
            }
        }
    };
    auto commit = [&](int buf) {
        float* As = lds + buf * 2 * PATCH;
        float* Bs = As + PATCH;
#pragma unroll
        for (int k = 0; k < STG; ++k) {
            const int idx = tid + k * 256;
            if (idx < PPH * PPW * 16) {
                if constexpr (DEFER) {
                    const float m = ma[k];
                    *reinterpret_cast<float4*>(As + (idx >> 4) * WS + 4 * (idx & 15)) =
                        make_float4(ra[k].x * (rs[k].x * m), ra[k].y * (rs[k].y * m), ra[k].z * (rs[k].z * m), ra[k].w * (rs[k].w * m));
                    *reinterpret_cast<float4*>(Bs + (idx >> 4) * WS + 4 * (idx & 15)) =
                        make_float4(rb[k].x * mb[k], rb[k].y * mb[k], rb[k].z * mb[k], rb[k].w * mb[k]);
                } else {
                    *reinterpret_cast<float4*>(As + (idx >> 4) * WS + 4 * (idx & 15)) = ra[k];
                    *reinterpret_cast<float4*>(Bs + (idx >> 4) * WS + 4 * (idx & 15)) = rb[k];
                }
            }
        }
    };

    // per-lane LDS word offsets: everything except the tap shift and the buffer is a compile-time constant of k,
    // so each operand fetch is one ds_read_b32 with an immediate offset
    int ta[NT], tb[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        ta[t] = (p.tap[t].ady * PPW + p.tap[t].adx + h) * WS + 32 * wi + l31;
        tb[t] = PATCH + (p.tap[t].bdy * PPW + p.tap[t].bdx + h) * WS + 32 * wj + l31;
    }

    int cur = 0;
    if (u_begin < u_end) { fetch(u_begin); commit(0); }
    __syncthreads();
    for (int u = u_begin; u < u_end; ++u) {
        if (u + 1 < u_end) fetch(u + 1);
        const int boff = cur * 2 * PATCH;
        // ---- K loop over the positions of the tile, two per MFMA (lane half h takes position 2k + h).
        // Operands of step k+1 are fetched from LDS before the MFMAs of step k are issued (register double buffer;
        // the sched_group_barriers pin that order), so the matrix pipe never waits for a ds_read.
        constexpr int KS = TPH * TPW / 2;
        float av[2][NT], bv[2][NT];
        auto ldk = [&](int k, int slot) {
            const int posoff = ((((2 * k) / TPW) + 1) * PPW + ((2 * k) % TPW) + 1) * WS;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (SHARE != 2 || t == 0) av[slot][t] = lds[boff + ta[t] + posoff];
                if (SHARE != 1 || t == 0) bv[slot][t] = lds[boff + tb[t] + posoff];
            }
        };
        ldk(0, 0);
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            if (k + 1 < KS) ldk(k + 1, (k + 1) & 1);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k & 1][SHARE == 2 ? 0 : t], bv[k & 1][SHARE == 1 ? 0 : t],
                                                              acc[t], 0, 0, 0);
        }
        if (u + 1 < u_end) commit(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // ---- store the slab: C/D layout row = (r&3) + 8*(r>>2) + 4*h (ci), col = lane&31 (co)
    float* slab = p.slabs + (size_t)ks * p.ntaps * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + 32 * wi + (r & 3) + 8 * (r >> 2) + 4 * h, co = co0 + 32 * wj + l31;
            if (ci < p.Cin && co < p.Cout) slab[((size_t)t * p.Cin + ci) * p.Cout + co] = acc[t][r];
        }
}

// dW[co][ci][tap] = sum_ks slab[ks][tap(widx)][ci][co]  -  W[co][ci][tap] * sum_b dd[b][co] d[b][co]^3 s[b][ci]^2
// thread = one (tap, ci, co) element, co fastest (coalesced slab reads); the slab loop is 8-way unrolled with
// independent partial sums (fixed order -> deterministic).  The first version looped over the taps inside a thread:
// the 128-channel layers (ksplit 128) had 64 blocks of 1152 dependent loads each — 2.7 ms per tuned fitting step.
// (Routing the stores through an LDS transpose so that they follow the parameter layout was measured: no gain, the
// slab reads bound this kernel.)
struct WTaps9 { WTap t[9]; };

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ slabs, const float* __restrict__ weight,
                                                           const float* __restrict__ dd, const float* __restrict__ dcoef,
                                                           const float* __restrict__ styles, float* __restrict__ dW,
                                                           int ksplit, int ntaps, int Cin, int Cout, int B, int wtaps,
                                                           const WTaps9 taps, int accumulate, int dds) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // over Cin * Cout, co fastest
    if (idx >= Cin * Cout) return;
    const int t = blockIdx.y;
    const int co = idx % Cout, ci = idx / Cout;
    float dem = 0.f;
    if (dd)
        for (int b = 0; b < B; ++b) {
            const float d = dcoef[(size_t)b * Cout + co], sv = styles[(size_t)b * Cin + ci];
            dem += dd[(size_t)b * dds + co] * d * d * d * sv * sv;
        }
    const float* src = slabs + ((size_t)t * Cin + ci) * Cout + co;
    const size_t kstride = (size_t)ntaps * Cin * Cout;
    float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= ksplit; k += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) part[u] += src[(size_t)(k + u) * kstride];
    }
    for (; k < ksplit; ++k) part[0] += src[(size_t)k * kstride];
    const float acc = ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
    const size_t wi = ((size_t)co * Cin + ci) * wtaps + taps.t[t].widx;
    const float v = acc - weight[wi] * dem;
    dW[wi] = accumulate ? dW[wi] + v : v;
}

// Tiled reducer (round 4): the reducer above stores dW[(co * Cin + ci) * wtaps + tap] from threads that are consecutive in co —
// 4-byte writes 18 KB apart (and the same gather on `weight`, also when dd is null and the product is 0): the 512 x 512 layers
// took 69 us for 38 MB of slab reads.  Here a block owns 32 co x 8 ci for ALL taps of the launch(es): slab reads stay coalesced
// in co, the sums go through LDS and leave as contiguous runs of 8 ci x wtaps floats per co (the parameter layout), and
// `weight` is only touched when the demodulation term exists.  Each tap names its own slab region, so the four parity
// launches of the up-sampling mode are reduced by ONE launch over all nine taps.
struct RedTap { long long base; long long kstride; int widx; };     // slab of split k at slabs + base + k * kstride + ci * Cout + co
struct RedTaps { RedTap t[9]; };

// V = floats per thread along co.  (Round 5 measured V = 4 — 16-byte reads, 8 ci x 128 co per block — at 92 us against 28: a
// quarter of the blocks, and the output phase with its per-element demodulation loads is latency-bound.  The split-bf16
// kernels no longer use this reducer: wgrad_sum_final_kernel below.)
template <int V>
__global__ void __launch_bounds__(256) wgrad_reduce_tiled_kernel(const float* __restrict__ slabs, const float* __restrict__ weight,
                                                                 const float* __restrict__ dd, const float* __restrict__ dcoef,
                                                                 const float* __restrict__ styles, float* __restrict__ dW,
                                                                 int ksplit, int ntaps, int Cin, int Cout, int B, int wtaps,
                                                                 const RedTaps taps, int accumulate, int dds) {
    constexpr int CW = 32 * V;
    __shared__ float tile[CW][8 * 9 + 1];                       // [co][ci * ntaps + tap slot]
    typedef float vec __attribute__((ext_vector_type(V)));
    const int tid = threadIdx.x, col = tid & 31, row = tid >> 5;
    const int ci0 = blockIdx.x * 8, co0 = blockIdx.y * CW;
    const int ci = ci0 + row, co = co0 + V * col;
    for (int t = 0; t < ntaps; ++t) {
        const float* src = slabs + taps.t[t].base + (size_t)ci * Cout + co;
        const long long ks = taps.t[t].kstride;
        vec part[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) part[u] = vec(0.f);
        int k = 0;
        for (; k + 4 <= ksplit; k += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) part[u] += *reinterpret_cast<const vec*>(src + (size_t)(k + u) * ks);
        }
        for (; k < ksplit; ++k) part[0] += *reinterpret_cast<const vec*>(src + (size_t)k * ks);
        const vec sum = (part[0] + part[1]) + (part[2] + part[3]);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            tile[V * col + v][row * ntaps + t] = sum[v];
        }
    }
    __syncthreads();
    // out: for each co a run of 8 ci x ntaps values; consecutive threads -> consecutive (ci, tap slot) of one co
    const int per_co = 8 * ntaps;
    for (int e = tid; e < CW * per_co; e += 256) {
        const int c = e / per_co, rem = e - c * per_co;
        const int r = rem / ntaps, t = rem - r * ntaps;
        const size_t wi = ((size_t)(co0 + c) * Cin + ci0 + r) * wtaps + taps.t[t].widx;
        float v = tile[c][rem];
        if (dd) {
            float dem = 0.f;
            for (int b = 0; b < B; ++b) {
                const float d = dcoef[(size_t)b * Cout + co0 + c], sv = styles[(size_t)b * Cin + ci0 + r];
                dem += dd[(size_t)b * dds + co0 + c] * d * d * d * sv * sv;
            }
            v -= weight[wi] * dem;
        }
        dW[wi] = accumulate ? dW[wi] + v : v;        // (accumulate: dweight is the parameter's .grad slice — ABI 11)
    }
}

// Reducer of the split-bf16 kernels (round 5): their slabs already have the parameter layout [Cout][Cin][9], so the reduction is
// an elementwise sum of `ksplit` arrays (16-byte loads, eight in flight per thread) minus the demodulation term.
__global__ void __launch_bounds__(256) wgrad_sum_final_kernel(const float* __restrict__ slabs, const float* __restrict__ weight,
                                                              const float* __restrict__ dd, const float* __restrict__ dcoef,
                                                              const float* __restrict__ styles, float* __restrict__ dW,
                                                              int ksplit, int Cin, int Cout, int B, int accumulate, int dds) {
    const size_t n4 = (size_t)Cin * Cout * 9 / 4;            // (Cin, Cout multiples of 64)
    const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= n4) return;
    const float4* src = reinterpret_cast<const float4*>(slabs) + i4;
    float4 part[4] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    int k = 0;
    for (; k + 8 <= ksplit; k += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(k + u) * n4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            part[u & 3].x += v[u].x; part[u & 3].y += v[u].y; part[u & 3].z += v[u].z; part[u & 3].w += v[u].w;
        }
    }
    for (; k < ksplit; ++k) {
        const float4 v = src[(size_t)k * n4];
        part[0].x += v.x; part[0].y += v.y; part[0].z += v.z; part[0].w += v.w;
    }
    float out[4] = {(part[0].x + part[1].x) + (part[2].x + part[3].x), (part[0].y + part[1].y) + (part[2].y + part[3].y),
                    (part[0].z + part[1].z) + (part[2].z + part[3].z), (part[0].w + part[1].w) + (part[2].w + part[3].w)};
    float4* dst = reinterpret_cast<float4*>(dW) + i4;
    if (dd) {
        const float4 w4 = reinterpret_cast<const float4*>(weight)[i4];
        const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t pair = (i4 * 4 + j) / 9;            // co * Cin + ci
            const int co = (int)(pair / Cin), ci = (int)(pair - (size_t)co * Cin);
            float dem = 0.f;
            for (int b = 0; b < B; ++b) {
                const float d = dcoef[(size_t)b * Cout + co], sv = styles[(size_t)b * Cin + ci];
                dem += dd[(size_t)b * dds + co] * d * d * d * sv * sv;
            }
            out[j] -= wv[j] * dem;
        }
    }
    if (accumulate) {
        const float4 o = *dst;
        out[0] += o.x; out[1] += o.y; out[2] += o.z; out[3] += o.w;
    }
    *dst = make_float4(out[0], out[1], out[2], out[3]);
}

static int launch_wgrad_sum_final(const HfagpWgradArgs* a, hipStream_t s) {
    const size_t n4 = (size_t)a->Cin * a->Cout * 9 / 4;
    wgrad_sum_final_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, s>>>(a->workspace, a->weight, a->dd, a->dcoef, a->styles, a->dweight,
                                                                        a->ksplit, a->Cin, a->Cout, a->B, a->accumulate, a->dd_stride > 0 ? a->dd_stride : a->Cout);
    return check_launch("conv_wgrad/sum");
}

// reduce `ntaps` taps described by `taps` (tap slots in widx order give contiguous stores); falls back to the per-element reducer
// for shapes the tiles do not divide
static int launch_wgrad_reduce(const HfagpWgradArgs* a, const RedTaps& rt, int ntaps, int wtaps, hipStream_t s) {
    wgrad_reduce_tiled_kernel<1><<<dim3((unsigned)(a->Cin / 8), (unsigned)(a->Cout / 32)), 256, 0, s>>>(
        a->workspace, a->weight, a->dd, a->dcoef, a->styles, a->dweight, a->ksplit, ntaps, a->Cin, a->Cout, a->B, wtaps, rt, a->accumulate,
        a->dd_stride > 0 ? a->dd_stride : a->Cout);
    return check_launch("conv_wgrad/reduce");
}

// dA[i][k] = wgain * sum_b dstot[b][i] * w[b][k];  db[i] = sum_b dstot[b][i]   (accumulating)
__global__ void __launch_bounds__(256) affine_grad_kernel(const float* __restrict__ dstot, const float* __restrict__ w,
                                                          float* __restrict__ dA, float* __restrict__ db, int B, int Cin,
                                                          int w_dim, int w_stride, float wgain) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cin * w_dim) return;
    const int i = idx / w_dim, k = idx % w_dim;
    float acc = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) {
        const float d = dstot[(size_t)b * Cin + i];
        acc += d * w[(size_t)b * w_stride + k];
        sb += d;
    }
    dA[idx] += acc * wgain;
    if (k == 0) db[i] += sb;
}

// the same for every affine layer of a backward pass in ONE launch (the tuned step ran 26 of the above: 3 us each and a
// zero-filled staging buffer; here dA / db are the parameters' .grad slices themselves)
constexpr int kBatchMax = 32;
struct AffineGradBatch { HfagpAffineGradItem it[kBatchMax]; };
__global__ void __launch_bounds__(256) affine_grad_batch_kernel(const AffineGradBatch t) {
    const HfagpAffineGradItem& a = t.it[blockIdx.y];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.Cin * a.w_dim) return;
    const int i = idx / a.w_dim, k = idx % a.w_dim;
    float acc = 0.f, sb = 0.f;
    for (int b = 0; b < a.B; ++b) {
        const float d = a.dstot[(size_t)b * a.Cin + i];
        acc += d * a.w[(size_t)b * a.w_stride + k];
        sb += d;
    }
    a.dA[idx] += acc * (1.0f / sqrtf((float)a.w_dim));
    if (k == 0) a.db[i] += sb;
}

// bias and noise-strength gradients of the synthesis layers of one block from the reductions hfagp_pointwise_bwd left behind
// (rows 4 and 5 of sums [B][10][C]):  dbias[c] += sum_b sums[b][4][c],  dnoise += sum_{b,c} sums[b][5][c]; fixed order
struct BiasNoiseBatch { HfagpBiasNoiseGradItem it[kBatchMax]; };
__global__ void __launch_bounds__(256) bias_noise_grads_kernel(const BiasNoiseBatch t) {
    __shared__ float red[256];
    const HfagpBiasNoiseGradItem& a = t.it[blockIdx.x];
    float n5 = 0.f;
    for (int c = threadIdx.x; c < a.C; c += 256) {
        float s4 = 0.f;
        for (int b = 0; b < a.B; ++b) {
            s4 += a.sums[((size_t)b * 10 + 4) * a.C + c];
            n5 += a.sums[((size_t)b * 10 + 5) * a.C + c];
        }
        if (a.dbias) a.dbias[c] += s4;
    }
    red[threadIdx.x] = n5;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0 && a.dnoise) a.dnoise[0] += red[0];
}

// per-channel sums over all pixels of a channels-last tensor: out[c] += sum_{b,pix} g[b][pix][c] (deterministic 2-stage).
// Round 5: the tensor is walked as a FLAT array with consecutive threads on consecutive floats (the first version gave a thread
// one channel of 256 / C pixel lanes — for the 3-channel image gradient 3 active lanes per block and a final pass of 3 threads
// summing 512 partials each: 118 us); the grid stride is a multiple of C, so a thread's channel never changes; four
// independent partial sums per thread; fixed order everywhere.
__global__ void __launch_bounds__(256) channel_sum_kernel(const float* __restrict__ g, float* __restrict__ partial,
                                                          long long n_elems, int C) {
    __shared__ float red[256];
    const long long stride = (long long)gridDim.x * 256;                 // multiple of C (checked by the host entry point)
    const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    long long i = i0;
    for (; i + 3 * stride < n_elems; i += 4 * stride) {
        a0 += g[i]; a1 += g[i + stride]; a2 += g[i + 2 * stride]; a3 += g[i + 3 * stride];
    }
    for (; i < n_elems; i += stride) a0 += g[i];
    red[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (threadIdx.x < C) {
        const int first = (int)(((long long)blockIdx.x * 256) % C);      // channel of thread 0
        float s = 0.f;
        for (int j = (threadIdx.x - first + C) % C; j < 256; j += C) s += red[j];
        partial[(size_t)blockIdx.x * C + threadIdx.x] = s;
    }
}

// one wave per channel: lane j sums the partials j, j + 64, ..., then a fixed butterfly
__global__ void __launch_bounds__(64) channel_sum_final_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                               int nblocks, int C, int accumulate) {
    const int c = blockIdx.x;
    float s = 0.f;
    for (int q = threadIdx.x; q < nblocks; q += 64) s += partial[(size_t)q * C + c];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (threadIdx.x == 0) out[c] = accumulate ? out[c] + s : s;
}

}  // namespace hfagp

namespace hfagp {      // wgrad_bf16.hip
int launch_wgrad3x3_bf16(const HfagpWgradArgs* a, hipStream_t s);
int launch_wgrad_up_bf16(const HfagpWgradArgs* a, hipStream_t s);
}

using namespace hfagp;

template <int NT, int SHARE>
static int run_wgrad(WgradParams& p, const HfagpWgradArgs* a, hipStream_t s, bool defer_reduce = false) {
    p.ntaps = NT;
    const size_t lds = (size_t)4 * PPH * PPW * WS * sizeof(float);        // two buffers x (x patch + g patch)
    dim3 grid((a->Cin + WT - 1) / WT, (a->Cout + WT - 1) / WT, a->ksplit);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<NT, SHARE>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    wgrad_kernel<NT, SHARE><<<grid, 256, lds, s>>>(p);
    int rc = check_launch("conv_wgrad");
    if (rc != HFAGP_OK) return rc;
    if (defer_reduce) return HFAGP_OK;
    const int wtaps = a->mode == HFAGP_CONV1X1 ? 1 : 9;
    if (a->Cin % 8 == 0 && a->Cout % 32 == 0) {
        RedTaps rt;
        const long long plane = (long long)a->Cin * a->Cout;
        for (int t = 0; t < NT; ++t) rt.t[t] = RedTap{(long long)(p.slabs - a->workspace) + t * plane, NT * plane, p.tap[t].widx};
        return launch_wgrad_reduce(a, rt, NT, wtaps, s);
    }
    WTaps9 taps;
    for (int t = 0; t < 9; ++t) taps.t[t] = p.tap[t];
    const dim3 rgrid((unsigned)((a->Cin * a->Cout + 255) / 256), (unsigned)NT);
    wgrad_reduce_kernel<<<rgrid, 256, 0, s>>>(p.slabs, a->weight, a->dd, a->dcoef, a->styles, a->dweight, a->ksplit, NT,
                                              a->Cin, a->Cout, a->B, wtaps, taps, a->accumulate,
                                              a->dd_stride > 0 ? a->dd_stride : a->Cout);
    return check_launch("conv_wgrad/reduce");
}

extern "C" {

// split-K policy (ABI 11; the host used to carry a copy of it): one resident block per CU for the split-bf16 kernel (measured 5 %
// better than two rounds: half the slabs to write and reduce), two rounds for the fp32 one; never more slabs than position tiles
int32_t hfagp_wgrad_ksplit(const HfagpWgradArgs* a) {
    if (!a || a->B <= 0 || a->H <= 0 || a->W <= 0 || a->Cin <= 0 || a->Cout <= 0) return 1;
    const bool up = a->mode == HFAGP_CONVT3X3_UP2;
    const bool split16 = a->precision == HFAGP_PREC_BF16X3 && (a->mode == HFAGP_CONV3X3 || up) && a->Cout % 64 == 0 &&
                         (a->Cin % 64 == 0 || (up && a->Cin == 32));
    const int rows = split16 ? 4 : 2;                  // position tile of the kernel that will run: rows x 16
    const long long units = (long long)a->B * ((a->H + rows - 1) / rows) * ((a->W + 15) / 16);
    long long tiles = (long long)((a->Cin + 63) / 64) * ((a->Cout + 63) / 64);
    const long long want = (split16 ? 256 : 512) / (tiles > 0 ? tiles : 1);
    long long k = want < 1 ? 1 : want;
    if (k > units) k = units;
    if (k > 256) k = 256;
    return (int32_t)(k < 1 ? 1 : k);
}

size_t hfagp_wgrad_workspace_bytes(const HfagpWgradArgs* a) {
    if (!a || a->ksplit <= 0) return 0;
    // (up-sampling mode: the four parity launches write disjoint regions — 4 + 2 + 2 + 1 taps — and ONE reducer finishes them)
    const int ntaps = a->mode == HFAGP_CONV1X1 ? 1 : 9;
    return (size_t)a->ksplit * ntaps * a->Cin * a->Cout * sizeof(float);
}

int hfagp_conv_wgrad(const HfagpWgradArgs* a, void* stream) {
    HFAGP_REQUIRE(a && a->x && a->g && a->weight && a->dweight && a->workspace, HFAGP_EBADARG, "conv_wgrad: null pointer");
    HFAGP_REQUIRE(a->Cin % 4 == 0 && a->Cout % 4 == 0 && a->B > 0 && a->H > 0 && a->W > 0 && a->ksplit >= 1,
                  HFAGP_EUNSUPPORTED, "conv_wgrad: Cin=%d Cout=%d must be multiples of 4", a->Cin, a->Cout);
    HFAGP_REQUIRE(!a->dd || (a->dcoef && a->styles), HFAGP_EBADARG, "conv_wgrad: dd needs dcoef and styles");
    WgradParams p{};
    p.x = a->x; p.styles = a->styles; p.g = a->g; p.slabs = a->workspace;
    p.B = a->B; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.ksplit = a->ksplit;
    p.tiles_h = (a->H + TPH - 1) / TPH; p.tiles_w = (a->W + TPW - 1) / TPW;      // position grid = x positions
    hipStream_t s = (hipStream_t)stream;
    if (a->mode == HFAGP_CONV3X3) {
        p.gH = a->H; p.gW = a->W;
        for (int t = 0; t < 9; ++t) p.tap[t] = WTap{(signed char)(t / 3 - 1), (signed char)(t % 3 - 1), 0, 0, (signed char)t};
        if (a->precision == HFAGP_PREC_BF16X3 && a->Cin % 64 == 0 && a->Cout % 64 == 0) {
            // split-bf16 MFMA kernel (slabs in the parameter layout), then the elementwise reducer
            int rc = launch_wgrad3x3_bf16(a, s);
            if (rc != HFAGP_OK) return rc;
            return launch_wgrad_sum_final(a, s);
        }
        HFAGP_REQUIRE(a->precision == HFAGP_PREC_F32 || a->precision == HFAGP_PREC_BF16X3, HFAGP_EBADARG,
                      "conv_wgrad: precision %d (F32 or BF16X3)", a->precision);
        return run_wgrad<9, 1>(p, a, s);
    }
    if (a->mode == HFAGP_CONV1X1) {
        p.gH = a->H; p.gW = a->W;
        p.tap[0] = WTap{0, 0, 0, 0, 0};
        return run_wgrad<1, 1>(p, a, s);
    }
    if (a->mode == HFAGP_CONVT3X3_UP2) {
        // g = the four parity images [2][2][B][H+1][W+1][Cout] of the y_t gradient; tap (ti, tj) reads parity
        // (ti&1, tj&1) at shift (ti>>1, tj>>1).  One launch per parity image.
        p.gH = a->H + 1; p.gW = a->W + 1;
        const long long img = (long long)a->B * p.gH * p.gW * a->Cout;
        static const int taps_of[4][4] = {{0, 2, 6, 8}, {1, 7, -1, -1}, {3, 5, -1, -1}, {4, -1, -1, -1}};
        static const int ntaps_of[4] = {4, 2, 2, 1};
        if (a->precision == HFAGP_PREC_BF16X3 && (a->Cin % 64 == 0 || a->Cin == 32) && a->Cout % 64 == 0) {
            // split-bf16 MFMA kernel: all nine taps in ONE launch (slabs in the parameter layout), then the elementwise reducer
            int rc = launch_wgrad_up_bf16(a, s);
            if (rc != HFAGP_OK) return rc;
            return launch_wgrad_sum_final(a, s);
        }
        // exact fp32: one launch per parity image; one reducer for the four when the tiles divide the layer (parity ph writes its
        // own slab region)
        const bool one_reduce = a->Cin % 8 == 0 && a->Cout % 32 == 0;
        const long long plane = (long long)a->Cin * a->Cout;
        static const int first_tap[4] = {0, 4, 6, 8};             // slab planes in front of parity ph, per split
        RedTaps rt;
        for (int ph = 0; ph < 4; ++ph) {
            p.g = a->g + ph * img;
            const long long region = one_reduce ? (long long)a->ksplit * first_tap[ph] * plane : 0;
            p.slabs = a->workspace + region;
            for (int k = 0; k < ntaps_of[ph]; ++k)
                rt.t[first_tap[ph] + k] = RedTap{region + k * plane, ntaps_of[ph] * plane, taps_of[ph][k]};
            for (int k = 0; k < ntaps_of[ph]; ++k) {
                const int t = taps_of[ph][k], ti = t / 3, tj = t % 3;
                p.tap[k] = WTap{0, 0, (signed char)(ti >> 1), (signed char)(tj >> 1), (signed char)t};
            }
            const int rc = ntaps_of[ph] == 4 ? run_wgrad<4, 2>(p, a, s, one_reduce) : ntaps_of[ph] == 2 ? run_wgrad<2, 2>(p, a, s, one_reduce)
                                                                                                      : run_wgrad<1, 2>(p, a, s, one_reduce);
            if (rc != HFAGP_OK) return rc;
        }
        if (one_reduce) return launch_wgrad_reduce(a, rt, 9, 9, s);
        return HFAGP_OK;
    }
    set_error("conv_wgrad: unsupported mode %d", a->mode);
    return HFAGP_EBADARG;
}

int hfagp_affine_grad(const float* dstot, const float* w, float* dA, float* db, int32_t B, int32_t Cin, int32_t w_dim,
                      int32_t w_stride, void* stream) {
    HFAGP_REQUIRE(dstot && w && dA && db, HFAGP_EBADARG, "affine_grad: null pointer");
    const int n = Cin * w_dim;
    affine_grad_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(dstot, w, dA, db, B, Cin, w_dim, w_stride,
                                                                         1.0f / sqrtf((float)w_dim));
    return check_launch("affine_grad");
}

int hfagp_affine_grad_batch(const HfagpAffineGradItem* items, int32_t n, void* stream) {
    HFAGP_REQUIRE(items && n >= 1 && n <= kBatchMax, HFAGP_EBADARG, "affine_grad_batch: 1..%d items", kBatchMax);
    AffineGradBatch t;
    int most = 0;
    for (int i = 0; i < n; ++i) {
        const HfagpAffineGradItem& a = items[i];
        HFAGP_REQUIRE(a.dstot && a.w && a.dA && a.db && a.B > 0 && a.Cin > 0 && a.w_dim > 0, HFAGP_EBADARG,
                      "affine_grad_batch: item %d: null pointer / bad dims", i);
        t.it[i] = a;
        most = std::max(most, a.Cin * a.w_dim);
    }
    affine_grad_batch_kernel<<<dim3((unsigned)((most + 255) / 256), (unsigned)n), 256, 0, (hipStream_t)stream>>>(t);
    return check_launch("affine_grad_batch");
}

int hfagp_bias_noise_grads(const HfagpBiasNoiseGradItem* items, int32_t n, void* stream) {
    HFAGP_REQUIRE(items && n >= 1 && n <= kBatchMax, HFAGP_EBADARG, "bias_noise_grads: 1..%d items", kBatchMax);
    BiasNoiseBatch t;
    for (int i = 0; i < n; ++i) {
        HFAGP_REQUIRE(items[i].sums && items[i].B > 0 && items[i].C > 0, HFAGP_EBADARG, "bias_noise_grads: item %d", i);
        t.it[i] = items[i];
    }
    bias_noise_grads_kernel<<<(unsigned)n, 256, 0, (hipStream_t)stream>>>(t);
    return check_launch("bias_noise_grads");
}

int hfagp_channel_sum(const float* g, float* partial, float* out, int64_t npix, int32_t C, int32_t nblocks,
                      int32_t accumulate, void* stream) {
    HFAGP_REQUIRE(g && partial && out, HFAGP_EBADARG, "channel_sum: null pointer");
    HFAGP_REQUIRE(C >= 1 && C <= 256 && nblocks >= 1, HFAGP_EUNSUPPORTED, "channel_sum: C=%d (max 256)", C);
    HFAGP_REQUIRE(((long long)nblocks * 256) % C == 0, HFAGP_EBADARG,
                  "channel_sum: nblocks * 256 must be a multiple of C (nblocks=%d, C=%d): a thread keeps one channel", nblocks, C);
    hipStream_t s = (hipStream_t)stream;
    channel_sum_kernel<<<nblocks, 256, 0, s>>>(g, partial, (long long)npix * C, C);
    channel_sum_final_kernel<<<C, 64, 0, s>>>(partial, out, nblocks, C, accumulate);
    return check_launch("channel_sum");
}

}  // extern "C"

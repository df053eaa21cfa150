// Fused tri-plane ray marcher (gfx950): one launch from (planes, camera, uniforms, decoder
// weights) to the composited 32-channel feature image.
//
// Replaces, for the call generator.synthesis(...) of headnerf.py:112, EG3D's
//   RaySampler.forward -> ImportanceRenderer.forward { sample_stratified, sample_from_planes
//   (F.grid_sample), OSGDecoder, MipRayMarcher2, sample_importance/sample_pdf, unify_samples
//   (sort + gather), MipRayMarcher2 }
// which materialise ~3 GB of intermediates per 128^2-ray frame.
//
// Mapping to the hardware
//   * one wavefront owns one ray at a time; a workgroup is 4 independent wavefronts.
//   * 16-sample tiles: lane = 16*g + j  (j = sample in tile, g = channel octet).  The four lanes
//     {j, j+16, j+32, j+48} read the four 32-byte pieces of each 128-byte texel line
//     (channels-last planes), 12 lines per sample, and accumulate the bilinear/plane mean in
//     8 registers — which is exactly the B-operand image of v_mfma_f32_16x16x4_f32, so the decoder
//     MLP runs on the matrix core straight from the gather registers:
//         H^T[64 x 16] = W0[64 x 32] . F^T      (32 MFMA)   softplus on the C registers
//         O^T[32 x 16] = W1c[32 x 64] . H^T     (32 MFMA)   C registers of layer 1 are the B operand
//     (the K order of both products is permuted so no cross-lane movement is needed); sigma, the
//     33rd output, is a 16-FMA VALU dot + two cross-octet shuffles.
//   * per-ray state (depths, densities, 96 x 32 colours) lives in a 17 KB LDS window per wave; the
//     transmittance / CDF scans are wavefront shuffle scans; the 48+48 merge is a rank count.
//   * final colour = sum_j omega_j c_j with omega_j = (w_{j-1} + w_j)/2 — algebraically the
//     midpoint rule of MipRayMarcher2, without forming the 95 midpoint colours.
#include "common.h"

namespace hfagp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CS = 36;   // LDS colour row stride in floats (32 + pad: conflict-free b128 writes)

struct RayParams {
    HfagpRaymarchArgs a;
    float lin_step;      // (float(end) - float(start)) / (Sc - 1)   [torch.linspace, fp32]
    float delta;         // float( (end - start) / (Sc - 1) )        [python double -> fp32]
    float coord_scale;   // 2 / box_warp
    int total_rays;
};

// Waves of a workgroup are independent here; LDS hand-offs between lanes of ONE wave only need the
// compiler not to reorder the accesses (the LDS executes a wave's DS instructions in order).
#define WAVE_SYNC()                                            \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
    } while (0)

// Transcendentals on the hardware units (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp each).  The libm
// forms (expf / log1pf / IEEE division) cost ~55 VALU instructions per softplus and made the kernel
// VALU-bound (17 k VALU instructions per ray, rocprofv3 SQ_INSTS_VALU); these cost ~8.
//   softplus(x) = max(x, 0) + log(1 + exp(-|x|))   (argument of log in (1, 2]: abs error ~1e-7;
//                                                   equals x for x > 20 like torch's threshold form)
__device__ __forceinline__ float exp_f(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float log_f(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + log_f(1.f + exp_f(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + exp_f(-x)); }

// inclusive product scan across the 64 lanes
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float u = __shfl_up(v, o);
        if (lane >= o) v *= u;
    }
    return v;
}
__device__ __forceinline__ float wave_scan_add(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int NC, int NF>
struct WaveLds {
    static constexpr int SC = 16 * NC, SF = 16 * NF, S = SC + SF;
    float col[S * CS];
    float t[S], sig[S];        // by sample id: coarse 0..SC-1, fine SC..S-1
    float ts[S], ss[S];        // sorted by depth
    float om[S];               // colour weight by sample id
    int sid[S];                // sorted position -> sample id
    float cdf[SC], tmid[SC];
};

template <int NC, int NF>
__global__ void __launch_bounds__(256, 2) raymarch_kernel(const RayParams p) {
    using L = WaveLds<NC, NF>;
    constexpr int SC = L::SC, SF = L::SF, S = L::S;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    L& lds = reinterpret_cast<L*>(smem)[wave];
    const HfagpRaymarchArgs& a = p.a;
    const int j = lane & 15, g = lane >> 4;
    const int R = a.res * a.res;

    // ---- decoder weights as MFMA A-operand registers (effective weights: W * lr_mul / sqrt(fan_in))
    const float g0 = a.decoder_lr_mul * 0.17677669529663687f;   // 1/sqrt(32)
    const float g1 = a.decoder_lr_mul * 0.125f;                 // 1/sqrt(64)
    float w0a[4][8], b0c[4][4], wsig[4][4], w1a[2][16], b1c[2][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int t = 0; t < 8; ++t) w0a[mt][t] = a.dec_w0[(16 * mt + j) * 32 + 8 * g + t] * g0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            b0c[mt][r] = a.dec_b0[16 * mt + 4 * g + r] * a.decoder_lr_mul;
            wsig[mt][r] = a.dec_w1[16 * mt + 4 * g + r] * g1;
        }
    }
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                w1a[ot][mt * 4 + r] = a.dec_w1[(1 + 16 * ot + j) * 64 + 16 * mt + 4 * g + r] * g1;
#pragma unroll
        for (int r = 0; r < 4; ++r) b1c[ot][r] = a.dec_b1[1 + 16 * ot + 4 * g + r] * a.decoder_lr_mul;
    }
    const float bsig = a.dec_b1[0] * a.decoder_lr_mul;

    const float fW = (float)a.W, fH = (float)a.H;
    const float sclx = fW * 0.5f, scly = fH * 0.5f;   // ATen CPU grid_sampler: (g + 1) * (size / 2) - 0.5

    for (int ray = blockIdx.x * 4 + wave; ray < p.total_rays; ray += gridDim.x * 4) {
        const int b = ray / R, rr = ray % R;
        const int pi = rr / a.res, pj = rr % a.res;
        // ---- ray generation (RaySampler.forward); wave-uniform
        const float* M = a.cam2world + b * 16;
        const float* K = a.intrinsics + b * 9;
        const float fx = K[0], sk = K[1], cx = K[2], fy = K[4], cy = K[5];
        const float inv_res = 1.0f / (float)a.res, half_res = 0.5f / (float)a.res;
        const float xc = __fadd_rn(__fmul_rn((float)pj, inv_res), half_res);
        const float yc = __fadd_rn(__fmul_rn((float)pi, inv_res), half_res);
        const float xl = __fdiv_rn(__fsub_rn(__fadd_rn(__fsub_rn(xc, cx), __fdiv_rn(__fmul_rn(cy, sk), fy)),
                                             __fdiv_rn(__fmul_rn(sk, yc), fy)), fx);
        const float yl = __fdiv_rn(__fsub_rn(yc, cy), fy);
        float o3[3], d3[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float wv = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(M[4 * k], xl), __fmul_rn(M[4 * k + 1], yl)),
                                                 M[4 * k + 2]), M[4 * k + 3]);
            o3[k] = M[4 * k + 3];
            d3[k] = __fsub_rn(wv, o3[k]);
        }
        {
            const float nrm = fmaxf(__fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(d3[0], d3[0]), __fmul_rn(d3[1], d3[1])),
                                                         __fmul_rn(d3[2], d3[2]))), 1e-12f);
            d3[0] = __fdiv_rn(d3[0], nrm); d3[1] = __fdiv_rn(d3[1], nrm); d3[2] = __fdiv_rn(d3[2], nrm);
        }

        // ---- stratified depths: torch.linspace(start, end, SC)[s] + u * delta
        if (lane < SC) {
            const float fs = (float)a.ray_start, fe = (float)a.ray_end;
            const float lin = lane < SC / 2 ? __fadd_rn(fs, __fmul_rn(p.lin_step, (float)lane))
                                            : __fsub_rn(fe, __fmul_rn(p.lin_step, (float)(SC - 1 - lane)));
            const float u = a.u_strat[(size_t)ray * SC + lane];
            lds.t[lane] = __fadd_rn(lin, __fmul_rn(u, p.delta));
        }
        WAVE_SYNC();

        // ---- gather + decoder for one 16-sample tile starting at sample id `s0`
        auto eval_tile = [&](int s0) {
            const int s = s0 + j;
            const float tz = lds.t[s];
            float q[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) q[k] = __fmul_rn(p.coord_scale, __fadd_rn(o3[k], __fmul_rn(tz, d3[k])));
            // plane projections: (x,y), (x,z), (z,x) [eg3d original] or (z,y) [fixed]
            const float gxs[3] = {q[0], q[0], q[2]};
            const float gys[3] = {q[1], q[2], a.plane_axes == 0 ? q[0] : q[1]};
            float f[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) f[c] = 0.f;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const float ix = __fsub_rn(__fmul_rn(__fadd_rn(gxs[pl], 1.f), sclx), 0.5f);
                const float iy = __fsub_rn(__fmul_rn(__fadd_rn(gys[pl], 1.f), scly), 0.5f);
                const float fx0 = floorf(ix), fy0 = floorf(iy);
                const float we = __fsub_rn(ix, fx0), ww = __fsub_rn(1.f, we);
                const float ws_ = __fsub_rn(iy, fy0), wn = __fsub_rn(1.f, ws_);
                // clamp before the int conversion so far-away coordinates stay defined
                const int x0 = (int)fminf(fmaxf(fx0, -2.f), fW + 1.f), y0 = (int)fminf(fmaxf(fy0, -2.f), fH + 1.f);
                const int x1 = x0 + 1, y1 = y0 + 1;
                const bool vx0 = x0 >= 0 && x0 < a.W, vx1 = x1 >= 0 && x1 < a.W;
                const bool vy0 = y0 >= 0 && y0 < a.H, vy1 = y1 >= 0 && y1 < a.H;
                const int cx0 = min(max(x0, 0), a.W - 1), cx1 = min(max(x1, 0), a.W - 1);
                const int cy0 = min(max(y0, 0), a.H - 1), cy1 = min(max(y1, 0), a.H - 1);
                const float w_nw = (vx0 && vy0) ? __fmul_rn(wn, ww) : 0.f;
                const float w_ne = (vx1 && vy0) ? __fmul_rn(wn, we) : 0.f;
                const float w_sw = (vx0 && vy1) ? __fmul_rn(ws_, ww) : 0.f;
                const float w_se = (vx1 && vy1) ? __fmul_rn(ws_, we) : 0.f;
                const float* base = a.planes + ((size_t)(b * 3 + pl) * a.H * a.W) * 32 + 8 * g;
                const float4* p_nw = reinterpret_cast<const float4*>(base + ((size_t)cy0 * a.W + cx0) * 32);
                const float4* p_ne = reinterpret_cast<const float4*>(base + ((size_t)cy0 * a.W + cx1) * 32);
                const float4* p_sw = reinterpret_cast<const float4*>(base + ((size_t)cy1 * a.W + cx0) * 32);
                const float4* p_se = reinterpret_cast<const float4*>(base + ((size_t)cy1 * a.W + cx1) * 32);
                const float4 nw0 = p_nw[0], nw1 = p_nw[1], ne0 = p_ne[0], ne1 = p_ne[1];
                const float4 sw0 = p_sw[0], sw1 = p_sw[1], se0 = p_se[0], se1 = p_se[1];
                const float v[4][8] = {{nw0.x, nw0.y, nw0.z, nw0.w, nw1.x, nw1.y, nw1.z, nw1.w},
                                       {ne0.x, ne0.y, ne0.z, ne0.w, ne1.x, ne1.y, ne1.z, ne1.w},
                                       {sw0.x, sw0.y, sw0.z, sw0.w, sw1.x, sw1.y, sw1.z, sw1.w},
                                       {se0.x, se0.y, se0.z, se0.w, se1.x, se1.y, se1.z, se1.w}};
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float acc = v[0][c] * w_nw;
                    acc = fmaf(v[1][c], w_ne, acc);
                    acc = fmaf(v[2][c], w_sw, acc);
                    acc = fmaf(v[3][c], w_se, acc);
                    f[c] += acc;
                }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) f[c] *= 0.3333333333333333f;   // mean over the 3 planes

            // layer 1: H^T = W0 . F^T + b0
            f32x4 h[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                h[mt] = f32x4{b0c[mt][0], b0c[mt][1], b0c[mt][2], b0c[mt][3]};
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    h[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0a[mt][t], f[t], h[mt], 0, 0, 0);
            }
            float sg = 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    h[mt][r] = softplus_f(h[mt][r]);
                    sg = fmaf(h[mt][r], wsig[mt][r], sg);
                }
            sg += __shfl_xor(sg, 16);
            sg += __shfl_xor(sg, 32);
            if (g == 0) lds.sig[s] = sg + bsig;
            // layer 2 (colour rows 1..32)
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                f32x4 o = f32x4{b1c[ot][0], b1c[ot][1], b1c[ot][2], b1c[ot][3]};
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        o = __builtin_amdgcn_mfma_f32_16x16x4f32(w1a[ot][mt * 4 + r], h[mt][r], o, 0, 0, 0);
                float4 cv;
                cv.x = sigmoid_f(o[0]) * 1.002f - 0.001f;
                cv.y = sigmoid_f(o[1]) * 1.002f - 0.001f;
                cv.z = sigmoid_f(o[2]) * 1.002f - 0.001f;
                cv.w = sigmoid_f(o[3]) * 1.002f - 0.001f;
                *reinterpret_cast<float4*>(&lds.col[s * CS + 16 * ot + 4 * g]) = cv;
            }
        };

        // ---- coarse pass
#pragma unroll 1
        for (int tile = 0; tile < NC; ++tile) eval_tile(16 * tile);
        WAVE_SYNC();

        // ---- coarse compositing weights (MipRayMarcher2) -> importance depths (sample_pdf)
        {
            float w = 0.f, sh = 1.f;
            const bool mid = lane < SC - 1;
            if (mid) {
                const float t0 = lds.t[lane], t1 = lds.t[lane + 1];
                const float dm = softplus_f((lds.sig[lane] + lds.sig[lane + 1]) * 0.5f - 1.f);
                const float alpha = 1.f - exp_f(-(dm * (t1 - t0)));
                sh = 1.f - alpha + 1e-10f;
                w = alpha;
                lds.tmid[lane] = 0.5f * (t0 + t1);
            }
            const float incl = wave_scan_mul(sh, lane);
            float T = __shfl_up(incl, 1);
            if (lane == 0) T = 1.f;
            w = mid ? w * T : -INFINITY;                   // lanes >= SC-1 act as the -inf padding
            float wp = __shfl_up(w, 1);
            if (lane == 0) wp = -INFINITY;
            const float m = fmaxf(wp, w);                  // max_pool1d(k=2, s=1, pad=1): SC values
            const float mn = __shfl_down(m, 1);
            const float sm = (m + mn) * 0.5f + 0.01f;      // avg_pool1d(k=2, s=1) + 0.01: lanes 0..SC-2
            const bool inpdf = lane >= 1 && lane <= SC - 3; // weights[:, 1:-1]
            const float pw = inpdf ? sm + 1e-5f : 0.f;
            const float tot = wave_sum(pw);
            const float pdf = inpdf ? pw / tot : 0.f;
            const float c = wave_scan_add(pdf, lane);
            if (lane <= SC - 3) lds.cdf[lane] = lane == 0 ? 0.f : c;   // SC-2 entries
        }
        WAVE_SYNC();
        if (lane < SF) {
            const float u = a.u_imp[(size_t)ray * SF + lane];
            int inds = 0;
            for (int k = 0; k < SC - 2; ++k) inds += lds.cdf[k] <= u ? 1 : 0;   // searchsorted(right=True)
            const int below = max(inds - 1, 0), above = min(inds, SC - 3);
            const float c0 = lds.cdf[below], c1 = lds.cdf[above];
            const float b0 = lds.tmid[below], b1 = lds.tmid[above];
            float den = c1 - c0;
            if (den < 1e-5f) den = 1.f;
            lds.t[SC + lane] = b0 + (u - c0) / den * (b1 - b0);
        }
        WAVE_SYNC();

        // ---- fine pass
#pragma unroll 1
        for (int tile = 0; tile < NF; ++tile) eval_tile(SC + 16 * tile);
        WAVE_SYNC();

        // ---- merge: rank of every sample in the union (coarse is already ascending)
        {
            const bool hc = lane < SC, hf = lane < SF;
            const float tc = hc ? lds.t[lane] : 0.f;
            const float tf = hf ? lds.t[SC + lane] : 0.f;
            int rc = lane, rf = 0;
            for (int k = 0; k < SF; ++k) {
                const float x = lds.t[SC + k];
                rc += x < tc ? 1 : 0;
                rf += (x < tf || (x == tf && k < lane)) ? 1 : 0;
            }
            for (int k = 0; k < SC; ++k) rf += lds.t[k] <= tf ? 1 : 0;
            if (hc) { lds.ts[rc] = tc; lds.ss[rc] = lds.sig[lane]; lds.sid[rc] = lane; }
            if (hf) { lds.ts[rf] = tf; lds.ss[rf] = lds.sig[SC + lane]; lds.sid[rf] = SC + lane; }
        }
        WAVE_SYNC();

        // ---- final compositing over the S-1 midpoints (two per lane)
        float wsum, dsum;
        {
            float al[2], sh[2], tm[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int e = lane + 64 * k;
                al[k] = 0.f; sh[k] = 1.f; tm[k] = 0.f;
                if (e < S - 1) {
                    const float t0 = lds.ts[e], t1 = lds.ts[e + 1];
                    const float dm = softplus_f((lds.ss[e] + lds.ss[e + 1]) * 0.5f - 1.f);
                    al[k] = 1.f - exp_f(-(dm * (t1 - t0)));
                    sh[k] = 1.f - al[k] + 1e-10f;
                    tm[k] = 0.5f * (t0 + t1);
                }
            }
            const float i0 = wave_scan_mul(sh[0], lane);
            const float tot0 = __shfl(i0, 63);
            float T0 = __shfl_up(i0, 1);
            if (lane == 0) T0 = 1.f;
            const float i1 = wave_scan_mul(sh[1], lane);
            float T1 = __shfl_up(i1, 1);
            if (lane == 0) T1 = 1.f;
            T1 *= tot0;
            const float w0 = al[0] * T0, w1 = al[1] * T1;   // zero beyond S-2
            wsum = wave_sum(w0 + w1);
            dsum = wave_sum(w0 * tm[0] + w1 * tm[1]);
            // omega_r = (w_{r-1} + w_r) / 2 for sorted position r
            float p0 = __shfl_up(w0, 1);
            if (lane == 0) p0 = 0.f;
            float p1 = __shfl_up(w1, 1);
            const float w0_63 = __shfl(w0, 63);
            if (lane == 0) p1 = w0_63;
            if (lane < S) lds.om[lds.sid[lane]] = 0.5f * (p0 + w0);
            if (lane + 64 < S) lds.om[lds.sid[lane + 64]] = 0.5f * (p1 + w1);
        }
        WAVE_SYNC();

        // ---- colour: rgb[c] = sum_s omega_s * col[s][c]   (two half-ranges of samples per channel)
        {
            const int c = lane & 31, hf = lane >> 5;
            float acc = 0.f;
            for (int s = hf * (S / 2); s < (hf + 1) * (S / 2); ++s) acc = fmaf(lds.om[s], lds.col[s * CS + c], acc);
            acc += __shfl_xor(acc, 32);
            if (a.white_back) acc = acc + 1.f - wsum;
            if (lane < 32) a.feat[(size_t)ray * 32 + c] = acc * 2.f - 1.f;
        }
        if (lane == 0) {
            float dep = dsum / wsum;
            if (dep != dep) dep = INFINITY;
            a.depth[ray] = dep;
            a.wsum[ray] = wsum;
            a.tminmax[2 * (size_t)ray] = lds.ts[0];
            a.tminmax[2 * (size_t)ray + 1] = lds.ts[S - 1];
        }
        WAVE_SYNC();
    }
}

template <int NC, int NF>
static int launch(const RayParams& p, hipStream_t s) {
    const size_t lds = 4 * sizeof(WaveLds<NC, NF>);
    int blocks = (p.total_rays + 3) / 4;
    const int cap = kNumCU * 2 * 4;          // 2 resident workgroups per CU, a few rounds each
    if (blocks > cap) blocks = cap;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&raymarch_kernel<NC, NF>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("raymarch: cannot raise dynamic LDS to %zu bytes: %s", lds, hipGetErrorString(e));
            return HFAGP_ELAUNCH;
        }
    }
    raymarch_kernel<NC, NF><<<blocks, 256, lds, s>>>(p);
    return check_launch("raymarch_fwd");
}

}  // namespace hfagp

using namespace hfagp;

extern "C" int hfagp_raymarch_fwd(const HfagpRaymarchArgs* a, void* stream) {
    HFAGP_REQUIRE(a && a->planes && a->cam2world && a->intrinsics && a->u_strat && a->u_imp && a->dec_w0 &&
                      a->dec_b0 && a->dec_w1 && a->dec_b1 && a->feat && a->depth && a->wsum && a->tminmax,
                  HFAGP_EBADARG, "raymarch_fwd: null pointer");
    HFAGP_REQUIRE(a->B > 0 && a->H > 1 && a->W > 1 && a->res > 0, HFAGP_EBADARG, "raymarch_fwd: bad dims");
    HFAGP_REQUIRE(a->box_warp > 0.f && a->ray_end > a->ray_start, HFAGP_EBADARG, "raymarch_fwd: bad ray range");
    RayParams p;
    p.a = *a;
    p.lin_step = ((float)a->ray_end - (float)a->ray_start) / (float)(a->Sc - 1);
    p.delta = (float)((a->ray_end - a->ray_start) / (double)(a->Sc - 1));
    p.coord_scale = (float)(2.0 / (double)a->box_warp);
    const long long total = (long long)a->B * a->res * a->res;
    HFAGP_REQUIRE(total < (1ll << 31), HFAGP_EUNSUPPORTED, "raymarch_fwd: too many rays");
    p.total_rays = (int)total;
    hipStream_t s = (hipStream_t)stream;
    if (a->Sc == 48 && a->Sf == 48) return launch<3, 3>(p, s);
    if (a->Sc == 32 && a->Sf == 32) return launch<2, 2>(p, s);
    if (a->Sc == 16 && a->Sf == 16) return launch<1, 1>(p, s);
    set_error("raymarch_fwd: unsupported sample counts Sc=%d Sf=%d (supported: 16+16, 32+32, 48+48)", a->Sc, a->Sf);
    return HFAGP_EUNSUPPORTED;
}

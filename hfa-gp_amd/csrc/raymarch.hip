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
#include <type_traits>
#include "raymarch_common.h"

namespace hfagp {

template <int NC, int NF>
struct WaveLds {
    static constexpr int SC = 16 * NC, SF = 16 * NF, S = SC + SF;
    float col[S * CS];
    float t[S], sig[S];        // by sample id: coarse 0..SC-1, fine SC..S-1
    float ts[S], ss[S];        // sorted by depth
    float om[S];               // colour weight by sample id
    int sid[S];                // sorted position -> sample id
    float cdf[SC], tmid[SC];
    float pp[S];               // backward: g . colour per sample
    float g2[32];              // backward: 2 * dL/dfeat of this ray
};

// GRADS = false: the forward renderer.  GRADS = true: first half of the backward pass — the same forward
// per ray, then the compositing adjoint, emitting one record (depth, omega, d sigma) per sample for
// raymarch_bwd_tiles_kernel (raymarch_bwd.hip).
// DEC16: the decoder MLP on the 16-bit matrix pipe with split fp16 operands (raymarch_common.h decoder_fwd16; needs
// HfagpRaymarchArgs::planes_absmax) instead of the exact fp32 matrix instructions — which were 0.9 of the 2.0 ms of an
// 8-frame launch.
// FROM_STATE (GRADS only): the per-sample colours, densities, depths and sort order of every ray are READ from
// HfagpRaymarchArgs::state, where the forward call of the same step left them (13.4 KB per ray), instead of being recomputed
// — no gather, no decoder: the compositing adjoint alone.
template <int NC, int NF, bool GRADS, bool DEC16, bool FROM_STATE = false>
__global__ void __launch_bounds__(256, 2) raymarch_kernel(const RayParams p) {
    static_assert(!FROM_STATE || GRADS, "the saved state is consumed by the backward pass");
    using L = WaveLds<NC, NF>;
    constexpr int SC = L::SC, SF = L::SF, S = L::S;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    L& lds = reinterpret_cast<L*>(smem)[wave];
    const HfagpRaymarchArgs& a = p.a;
    const int j = lane & 15, g = lane >> 4;
    const int R = a.res * a.res;

    constexpr bool kL1Lds = DEC16 && !FROM_STATE;   // layer 1 of the 16-bit decoder from LDS (raymarch_common.h decoder_fwd16_l1)
    typename std::conditional<DEC16, Dec16Regs, DecoderRegs>::type dec;
    if constexpr (FROM_STATE) {
        (void)dec; (void)j; (void)g;
    } else if constexpr (DEC16) {
        DecoderRegs dec32;
        load_decoder(a, j, g, dec32);
        make_dec16(dec32, a.planes_absmax, lane, dec);
        if constexpr (kL1Lds) {                  // layer 1 of the decoder -> workgroup-shared LDS image (behind the four wave windows)
            if (wave == 0) store_dec16_l1(dec, reinterpret_cast<float*>(smem + 4 * sizeof(L)), lane);
            prescale_dec16_l1(dec);              // (what stays in registers moves to base 2 as well)
            __syncthreads();
        }
    } else {
        load_decoder(a, j, g, dec);
    }
    constexpr int kStateFloats = S * 35;        // per ray: col [S][32] | ts [S] | ss [S] | sid [S]

    const RaySchedule sch = ray_schedule(p.total_rays, wave);
    // The set-up of a ray — position in the sequence -> (frame, pixel) with five integer divisions, the camera ray with seven
    // IEEE divisions and a square root — is the same for all 64 lanes: computed per ray it was ~300 redundant instructions.
    // Instead lane l prepares the wave's ray number l of the next 64, and the ray loop picks its values with v_readlane.
    for (long long q0 = sch.begin; q0 < sch.end; q0 += 64ll * sch.stride) {      // (64-bit: the last step may pass 2^31)
    const int seq0 = __builtin_amdgcn_readfirstlane((int)q0);
    int b_l = 0, ray_l = 0;
    float o_l[3] = {0.f, 0.f, 0.f}, d_l[3] = {0.f, 0.f, 0.f};
    {
        const long long sl = (long long)seq0 + (long long)lane * sch.stride;
        if (sl < sch.end) {
            int pi, pj;
            ray_of((int)sl, a.res, b_l, pi, pj);
            ray_l = b_l * R + pi * a.res + pj;                 // index into the [B][R] tensors
            if constexpr (!FROM_STATE) ray_setup(a, b_l, pi, pj, o_l, d_l);
        }
    }
    const int nbatch = __builtin_amdgcn_readfirstlane(min(64, ((int)sch.end - seq0 + sch.stride - 1) / sch.stride));
    #pragma unroll 1
    for (int k = 0; k < nbatch; ++k) {
        const int b = __builtin_amdgcn_readlane(b_l, k), ray = __builtin_amdgcn_readlane(ray_l, k);    // wave-uniform -> scalar registers
        if constexpr (FROM_STATE) {
            const float* st = a.state + (size_t)ray * kStateFloats;
            // 16-byte loads, ALL of the ray's colours in flight at once (12 per lane at 48 + 48 samples): with 4-byte loads, four in
            // flight, a ray cost twelve memory round trips and this pass ran at a third of the rate the 440 MB of state allow
            {
                constexpr int NV = S * 8 / 64;
                f32x4 v[NV];
                const f32x4* st4 = reinterpret_cast<const f32x4*>(st);
#pragma unroll
                for (int k = 0; k < NV; ++k) v[k] = __builtin_nontemporal_load(st4 + lane + 64 * k);
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int i = lane + 64 * k;
                    *reinterpret_cast<f32x4*>(&lds.col[(i >> 3) * CS + 4 * (i & 7)]) = v[k];
                }
            }
            for (int i = lane; i < S; i += 64) {
                lds.ts[i] = st[S * 32 + i];
                lds.ss[i] = st[S * 33 + i];
                lds.sid[i] = __float_as_int(st[S * 34 + i]);
            }
        } else {
            float o3[3], d3[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                o3[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o_l[i]), k));
                d3[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d_l[i]), k));
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
            // The gather runs in a quad layout — lanes 4q..4q+3 read the four 32-B channel groups of the SAME texel line of
            // sample q — so that a load instruction touches 16 lines with 4 adjacent lanes each, not 64 lines with one lane
            // each (4x fewer tag look-ups in the texture addresser: 2.6 -> 2.0 ms per 8 frames; 8 lanes per line with two
            // samples per lane measured 2.3 ms).  The interpolated features then move to the MFMA layout (lane 16g + j <-
            // lane 4j + g).
            auto decode_tile = [&](int s0, float f[8]) {
                const int s = s0 + j;
                const int src = 4 * j + g;
    #pragma unroll
                for (int c = 0; c < 8; ++c) f[c] = __shfl(f[c], src);
                f32x4 o[2];
                float sigma;
                if constexpr (kL1Lds) {
                    decoder_fwd16_l1(dec, reinterpret_cast<const float*>(smem + 4 * sizeof(L)), lane, f, sigma, o);
                } else {
                    f32x4 h[4];
                    if constexpr (DEC16) decoder_fwd16<false>(dec, f, h, h, sigma, o);
                    else decoder_fwd<false>(dec, f, h, h, sigma, o);
                }
                if (g == 0) lds.sig[s] = sigma;
    #pragma unroll
                for (int ot = 0; ot < 2; ++ot) {
                    float4 cv;
                    if constexpr (kL1Lds) {          // logits in base 2
                        cv.x = sigmoid2_f(o[ot][0]) * 1.002f - 0.001f;
                        cv.y = sigmoid2_f(o[ot][1]) * 1.002f - 0.001f;
                        cv.z = sigmoid2_f(o[ot][2]) * 1.002f - 0.001f;
                        cv.w = sigmoid2_f(o[ot][3]) * 1.002f - 0.001f;
                    } else {
                        cv.x = sigmoid_f(o[ot][0]) * 1.002f - 0.001f;
                        cv.y = sigmoid_f(o[ot][1]) * 1.002f - 0.001f;
                        cv.z = sigmoid_f(o[ot][2]) * 1.002f - 0.001f;
                        cv.w = sigmoid_f(o[ot][3]) * 1.002f - 0.001f;
                    }
                    *reinterpret_cast<float4*>(&lds.col[s * CS + 16 * ot + 4 * g]) = cv;
                }
            };
            auto eval_tile = [&](int s0) {
                float f[8];
                PlaneTaps taps[3];
                sample_taps(p, o3, d3, lds.t[s0 + (lane >> 2)], taps);
                gather8(a, b, lane & 3, taps, f);
                decode_tile(s0, f);
            };
            // forward kernel on the 16-bit decoder: every tile's 24 texel loads in flight at once (the 64 registers come from
            // layer 1 of the decoder living in LDS).  Issuing the loads of tile t+1 BEFORE the decoder of tile t on top of that
            // needs 108 registers live across the decoder: 85 spilled, not pursued; the tap arithmetic of tile t+1 placed under
            // the load latency of tile t fits (238 registers) and measured 2 % SLOWER (profiles/r03_raymarch_valu.md).
            auto eval_pass = [&](int first, int ntiles) {
                if constexpr (kL1Lds) {
    #pragma unroll 1
                    for (int tile = 0; tile < ntiles; ++tile) {
                        float f[8];
                        {
                            TileLoads tl;
                            PlaneTaps taps[3];
                            // (priority: this wave's tap arithmetic and load issue go ahead of the other wave's decoder
                            // arithmetic on the SIMD, so its loads are in flight under that decoder: -1.7 %)
                            __builtin_amdgcn_s_setprio(3);
                            sample_taps(p, o3, d3, lds.t[first + 16 * tile + (lane >> 2)], taps);
                            tile_issue(a, b, lane & 3, taps, tl);
                            __builtin_amdgcn_s_setprio(0);
                            tile_reduce(tl, f);
                        }
                        decode_tile(first + 16 * tile, f);
                    }
                } else {
    #pragma unroll 1
                    for (int tile = 0; tile < ntiles; ++tile) eval_tile(first + 16 * tile);
                }
            };

            // ---- coarse pass
            eval_pass(0, NC);
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
                // searchsorted(right=True) = #{k < SC-2 : cdf[k] <= u}; the cdf is a running sum of non-negative terms, so a
                // bisection counts the same thing as the 46-step scan it replaces (6 LDS reads instead of 46)
                int inds = 0;
#pragma unroll
                for (int step = 32; step > 0; step >>= 1) {
                    const int probe = inds + step;
                    if (probe <= SC - 2 && lds.cdf[probe - 1] <= u) inds = probe;
                }
                const int below = max(inds - 1, 0), above = min(inds, SC - 3);
                const float c0 = lds.cdf[below], c1 = lds.cdf[above];
                const float b0 = lds.tmid[below], b1 = lds.tmid[above];
                float den = c1 - c0;
                if (den < 1e-5f) den = 1.f;
                lds.t[SC + lane] = b0 + (u - c0) / den * (b1 - b0);
            }
            WAVE_SYNC();

            // ---- fine pass
            eval_pass(SC, NF);
            WAVE_SYNC();

            // ---- merge: rank of every sample in the union, stable, coarse before fine on ties.  The coarse depths are already
            // ascending, so   rank(fine l)   = cf_l + #{fine k before l},   cf_l = #{coarse <= tf_l}   (bisection)
            //                 rank(coarse i) = i + #{fine l : cf_l <= i}                             (histogram of cf + scan)
            // and only "fine before fine" is a count over all SF values — strict comparisons, with the k < l tie-break taken
            // on a (wave-uniform, rare) second pass when two fine depths are EQUAL, which the duplicate ranks reveal.
            // 485 -> ~190 VALU instructions per ray.
            {
                static_assert(SC < 64 && SF <= 64 && SF % 4 == 0, "one lane per coarse / fine sample, SC + 1 histogram bins");
                const bool hc = lane < SC, hf = lane < SF;
                const float tc = hc ? lds.t[lane] : 0.f;
                const float tf = hf ? lds.t[SC + lane] : 0.f;
                int* bins = reinterpret_cast<int*>(lds.om);          // (om is written by the final compositing, after this)
                if (lane <= SC) bins[lane] = 0;
                if (lane < S) lds.sid[lane] = 0;                     // duplicate detector, indexed by rank
                if (lane + 64 < S) lds.sid[lane + 64] = 0;
                int cf = 0;
#pragma unroll
                for (int step = 32; step > 0; step >>= 1) {
                    const int probe = cf + step;
                    if (probe <= SC && lds.t[probe - 1] <= tf) cf = probe;
                }
                WAVE_SYNC();
                if (hf) atomicAdd(&bins[cf], 1);
                int lt = 0;
#pragma unroll
                for (int k = 0; k < SF; k += 4) {
                    const float4 x = *reinterpret_cast<const float4*>(&lds.t[SC + k]);      // broadcast reads
                    lt += (x.x < tf ? 1 : 0) + (x.y < tf ? 1 : 0) + (x.z < tf ? 1 : 0) + (x.w < tf ? 1 : 0);
                }
                WAVE_SYNC();
                const float below = wave_scan_add(lane <= SC ? (float)bins[lane] : 0.f, lane);   // #{fine : cf <= lane}: exact, <= SF
                const int rc = lane + (int)below;
                int rf = cf + lt;
                const int before = hf ? atomicAdd(&lds.sid[rf], 1) : 0;
                if (__builtin_amdgcn_ballot_w64(before != 0) != 0) {          // equal fine depths: the stable order
                    for (int k = 0; k < SF; ++k) rf += (k < lane && lds.t[SC + k] == tf) ? 1 : 0;
                }
                WAVE_SYNC();
                if (hc) { lds.ts[rc] = tc; lds.ss[rc] = lds.sig[lane]; lds.sid[rc] = lane; }
                if (hf) { lds.ts[rf] = tf; lds.ss[rf] = lds.sig[SC + lane]; lds.sid[rf] = SC + lane; }
            }
        }
        WAVE_SYNC();
        if constexpr (!GRADS) {
            if (a.state) {                          // forward of a step that will be differentiated: leave the state behind
                float* st = a.state + (size_t)ray * kStateFloats;
#pragma unroll 4
                for (int i = lane; i < S * 8; i += 64)           // 16 bytes per lane (the row stride CS keeps them aligned); streaming:
                    __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(&lds.col[(i >> 3) * CS + 4 * (i & 7)]),      // 13 KB per ray
                                                reinterpret_cast<f32x4*>(st) + i);                                          // must not evict the plane bands
                for (int i = lane; i < S; i += 64) {
                    st[S * 32 + i] = lds.ts[i];
                    st[S * 33 + i] = lds.ss[i];
                    st[S * 34 + i] = __int_as_float(lds.sid[i]);
                }
            }
        }

        // ---- backward only: P_j = sum_c 2 dL/dfeat[c] * colour_j[c]
        float gsum2 = 0.f;
        if constexpr (GRADS) {
            if (lane < 32) lds.g2[lane] = 2.f * p.g_feat[(size_t)ray * 32 + lane];
            WAVE_SYNC();
            gsum2 = wave_sum(lane < 32 ? lds.g2[lane] : 0.f);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int s = lane + 64 * k;
                if (s < S) {
                    float acc = 0.f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float4 cv = *reinterpret_cast<const float4*>(&lds.col[s * CS + 4 * q]);
                        const float4 gv = *reinterpret_cast<const float4*>(&lds.g2[4 * q]);
                        acc += cv.x * gv.x + cv.y * gv.y + cv.z * gv.z + cv.w * gv.w;
                    }
                    lds.pp[s] = acc;
                }
            }
            WAVE_SYNC();
        }

        // ---- final compositing over the S-1 midpoints (two per lane)
        float wsum, dsum;
        {
            float al[2], sh[2], tm[2], dl[2], sbar[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int e = lane + 64 * k;
                al[k] = 0.f; sh[k] = 1.f; tm[k] = 0.f; dl[k] = 0.f; sbar[k] = 0.f;
                if (e < S - 1) {
                    const float t0 = lds.ts[e], t1 = lds.ts[e + 1];
                    sbar[k] = (lds.ss[e] + lds.ss[e + 1]) * 0.5f - 1.f;
                    const float dm = softplus_f(sbar[k]);
                    dl[k] = t1 - t0;
                    al[k] = 1.f - exp_f(-(dm * dl[k]));
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
            if constexpr (GRADS) {
                // adjoint of the compositing.  rgb = sum_e w_e cbar_e (+ white_back term),
                //   G_e = g . cbar_e - wb * sum(g),   dL/dalpha_e = G_e T_e - (sum_{k>e} G_k w_k) / (1 - alpha_e + eps)
                //   dalpha/dsigma~ = delta (1 - alpha),  dsigma~/dsigmabar = sigmoid(sigmabar - 1)
                const float wb = a.white_back ? gsum2 : 0.f;
                float G[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int e = lane + 64 * k;
                    G[k] = e < S - 1 ? 0.5f * (lds.pp[lds.sid[e]] + lds.pp[lds.sid[e + 1]]) - wb : 0.f;
                }
                const float gw0 = G[0] * w0, gw1 = G[1] * w1;
                const float inc0 = wave_scan_add(gw0, lane), inc1 = wave_scan_add(gw1, lane);
                const float tot_a = __shfl(inc0, 63), tot = tot_a + __shfl(inc1, 63);
                const float suf0 = tot - inc0, suf1 = tot - (tot_a + inc1);
                const float da0 = G[0] * T0 - suf0 / sh[0], da1 = G[1] * T1 - suf1 / sh[1];
                float ds0 = da0 * dl[0] * (1.f - al[0]) * sigmoid_f(sbar[0]);
                float ds1 = da1 * dl[1] * (1.f - al[1]) * sigmoid_f(sbar[1]);
                if (lane >= S - 1) ds0 = 0.f;
                if (lane + 64 >= S - 1) ds1 = 0.f;
                float q0 = __shfl_up(ds0, 1);
                if (lane == 0) q0 = 0.f;
                float q1 = __shfl_up(ds1, 1);
                const float ds0_63 = __shfl(ds0, 63);
                if (lane == 0) q1 = ds0_63;
                // records in DEPTH order (sorted position, not sample id): a 16-sample tile of pass 2 is then a contiguous
                // piece of the ray, so consecutive samples land on the same or neighbouring texels — the importance samples
                // cluster at the surface — and the scatter's run merging folds them into one atomic per texel line
                float* rec = p.rec + (size_t)ray * S * 4;
                if (lane < S)
                    *reinterpret_cast<float4*>(rec + lane * 4) = make_float4(lds.ts[lane], 0.5f * (p0 + w0), 0.5f * (q0 + ds0), 0.f);
                if (lane + 64 < S)
                    *reinterpret_cast<float4*>(rec + (lane + 64) * 4) = make_float4(lds.ts[lane + 64], 0.5f * (p1 + w1), 0.5f * (q1 + ds1), 0.f);
            }
        }
        WAVE_SYNC();
        if constexpr (GRADS) continue;      // the forward outputs are not needed again

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
}

template <int NC, int NF, bool GRADS, bool DEC16, bool FROM_STATE = false>
static int launch(const RayParams& p, hipStream_t s) {
    const size_t lds = 4 * sizeof(WaveLds<NC, NF>) + ((DEC16 && !FROM_STATE) ? kDecL1Floats * sizeof(float) : 0);
    int blocks = (p.total_rays + 3) / 4;
    const int cap = kNumCU * 2 * 4;          // 2 resident workgroups per CU, a few rounds each
    if (blocks > cap) blocks = cap;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&raymarch_kernel<NC, NF, GRADS, DEC16, FROM_STATE>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) {
            set_error("raymarch: cannot raise dynamic LDS to %zu bytes: %s", lds, hipGetErrorString(e));
            return HFAGP_ELAUNCH;
        }
    }
    raymarch_kernel<NC, NF, GRADS, DEC16, FROM_STATE><<<blocks, 256, lds, s>>>(p);
    return check_launch(GRADS ? "raymarch_bwd/samples" : "raymarch_fwd");
}

template <bool GRADS, bool DEC16, bool FROM_STATE = false>
static int launch_n(const RayParams& p, hipStream_t s) {
    const int n = p.a.Sc / 16;
    return n == 3 ? launch<3, 3, GRADS, DEC16, FROM_STATE>(p, s) : n == 2 ? launch<2, 2, GRADS, DEC16, FROM_STATE>(p, s)
                                                                          : launch<1, 1, GRADS, DEC16, FROM_STATE>(p, s);
}

int launch_raymarch(const RayParams& p, bool grads, hipStream_t s) {
    const bool dec16 = p.a.planes_absmax != nullptr;
    if (grads && p.a.state) return launch_n<true, false, true>(p, s);        // (no decoder in this variant)
    if (grads) return dec16 ? launch_n<true, true>(p, s) : launch_n<true, false>(p, s);
    return dec16 ? launch_n<false, true>(p, s) : launch_n<false, false>(p, s);
}

}  // namespace hfagp

using namespace hfagp;

extern "C" int hfagp_raymarch_fwd(const HfagpRaymarchArgs* a, void* stream) {
    HFAGP_REQUIRE(a, HFAGP_EBADARG, "raymarch_fwd: null pointer");
    HFAGP_REQUIRE(a->feat && a->depth && a->wsum && a->tminmax, HFAGP_EBADARG, "raymarch_fwd: null pointer");
    RayParams p;
    const int rc = fill_ray_params(a, p, "raymarch_fwd");
    if (rc != HFAGP_OK) return rc;
    return launch_raymarch(p, false, (hipStream_t)stream);
}

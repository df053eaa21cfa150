// Backward of the fused ray marcher w.r.t. the tri-plane volume (gfx950).
//
//   pass 1 (raymarch_kernel<.., GRADS=true>, raymarch.hip): re-runs the forward per ray (both sampling
//           passes, merge, compositing) and its compositing adjoint; emits per sample (depth, omega, d sigma)
//           where dL/dcolour_j = omega_j * 2 dL/dfeat and dL/dsigma_j = d sigma.   18 MB / frame.
//   pass 2 (raymarch_bwd_tiles_kernel, here): one wavefront per 16-sample tile, no per-ray state:
//           gather -> decoder forward (recomputed) -> decoder backward on the matrix core
//             dH^T[64x16] = W1c^T . dO^T + wsig (x) dsigma      (32 MFMA; the C layout of dO is the B operand)
//             dF^T[32x16] = W0^T . (dH * sigmoid(Hpre))^T       (32 MFMA)
//           -> dF transposed through LDS so that each half-wave owns the 32 channels of ONE texel line ->
//           fp32 atomic adds of full 128-byte lines into d_planes (12 taps per sample).
//   pass 2 since round 5, whenever the caller provides HfagpRaymarchBwdArgs::rows_scratch: SORT + GATHER (raymarch_rows.hip) — the
//           samples are counting-sorted by (plane, column strip, texel row); raymarch_bwd_df_kernel<.., SORTED> below (generator
//           frozen) or raymarch_bwd_tiles_kernel<PG, .., SCATTER = false, WIDE> (tuned: with the decoder gradients) write dL/dF to
//           the sorted slots; raymarch_bwd_rows_kernel forms every row tile of d_planes as a dense product.  The scatter kernels
//           of this file (tiles: atomics; cols: LDS line cache) remain the fallback for shapes the sort does not take.
//   The importance depths carry no gradient (EG3D: no_grad + detach), neither do the camera / depths.
#include <algorithm>
#include <cstdlib>
#include "raymarch_common.h"

namespace hfagp {

struct TileLds {
    float df[16 * 32];        // dL/dfeature of the 16 samples, [sample][channel]
    __attribute__((aligned(16))) int   idx[12 * 16];       // texel index of every tap, [plane*4 + tap][sample]
    __attribute__((aligned(16))) float wgt[12 * 16];       // bilinear weight / 3
};

// decoder-weight gradients (PG): per-tile operand images for the sample-contracting products
//   dW1c[32x64] += dO[32x16] . SP^T[16x64]      dW0[64x32] += dHpre[64x16] . F[16x32]
struct GradLds {
    float sp[64 * 17], dp[64 * 17], dO[32 * 17], f[16 * 33];
};

struct DecGrads { float *w0, *b0, *w1, *b1; };   // [64][32], [64], [33][64], [33]; accumulated with atomics

// MIRROR: the projections of planes 1 and 2 are (x,z) and (z,x) on square planes (EG3D's original axes): their taps
// are the same texels with row/column swapped and bitwise the same weights, and dL/dfeature is shared by the three
// planes (the decoder sees their mean), so  d_planes[2][x][z] == d_planes[1][z][x]  exactly.  Plane 2 is then not
// scattered at all (a third of the atomics, whose issue rate bounds this kernel) and mirror_plane_kernel fills it.
// Waves per workgroup: 6 without the decoder gradients — the three weight images (43 KB) are shared by the block, so
// 6 waves need 64 KB and two workgroups = 3 waves per SIMD fit a CU (4-wave blocks: 57.6 KB each, 2 per SIMD);
// 4 with them (PG: 13 KB more per wave, one workgroup per CU).
template <bool PG> struct BwdWaves { static constexpr int value = PG ? 4 : 6; };

// DEC16: decoder forward (split fp16) and backward (split bf16) on the 16-bit matrix pipe (raymarch_common.h; needs
// HfagpRaymarchArgs::planes_absmax): 48 MFMAs of ~17 cycles per tile instead of 128 fp32 ones of 32.
// SCATTER = false (PG only): the decoder-parameter gradients alone — d planes comes from raymarch_bwd_cols_kernel (the
// generator-tuned step on mirrored square planes: the column kernel's LDS line cache scatters in 1.6 ms what this kernel's
// global atomics take 3.2 ms for; what is left here is gather + decoder forward / backward + the weight-gradient MFMAs).
// (the decoder-gradient-only variant with 8 waves per CU — the LDS would allow it without the scatter tables — measured
// 2.2 ms against 1.9 ms: 256 registers per wave instead of 512 spill)
// WIDE (PG, no scatter): eight waves per workgroup = two per SIMD at 256 registers, the weight images read from LDS per tile
// (a laundered lane offset keeps hipcc from hoisting ~170 registers of them out of the loop) and no software pipeline — against
// one wave per SIMD with 512 registers and the pipeline.
template <bool PG, bool SCATTER, bool WIDE = false> struct TileWaves { static constexpr int value = SCATTER ? BwdWaves<PG>::value : WIDE ? 8 : 4; };

template <int S, bool PG, bool MIRROR, bool DEC16, bool SCATTER = true, bool WIDE = false>
__global__ void __launch_bounds__((TileWaves<PG, SCATTER, WIDE>::value * 64), (PG ? 1 : 2))
raymarch_bwd_tiles_kernel(const RayParams p, float* __restrict__ d_planes, const DecGrads dg, const RowsOut ro) {
    // (SCATTER = false, PG = false: dL/dF to the sorted slots alone — the generator-frozen pass of the sort + gather form)
    static_assert(!WIDE || (PG && !SCATTER), "the wide form is the decoder-gradient pass");
    constexpr int NWB = TileWaves<PG, SCATTER, WIDE>::value, NTHB = NWB * 64;
    __shared__ TileLds lds_all[SCATTER ? NWB : 1];
    // (sized 1 float instead of one GradLds when the decoder gradients are off: 13 KB less -> 3 workgroups per CU)
    __shared__ __attribute__((aligned(16))) float glds_raw[PG ? NWB * sizeof(GradLds) / sizeof(float) : 1];
    // A operands of the two backward products, lane-linear ([step][lane]: conflict-free ds_read_b32), shared by
    // the 4 waves:  w1t[mt][ot*4+r][lane] = W1[1 + 16ot + 4g + r][16mt + j] * g1
    //               w0t[ft][mt*4+r][lane] = W0[16mt + 4g + r][16ft + j] * g0
    __shared__ float w1t[4 * 8 * 64];
    __shared__ float w0t[2 * 16 * 64];
    __shared__ float wfwd[kDecLdsRows * 64];       // forward A operands (DecoderRegs image)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    TileLds& lds = lds_all[SCATTER ? wave : 0];     // never dereferenced unless SCATTER
    const HfagpRaymarchArgs& a = p.a;
    const int j = lane & 15, g = lane >> 4;
    const int R = a.res * a.res;
    constexpr int NT = S / 16;

    if (wave == 0) {
        DecoderRegs dec;
        load_decoder(a, j, g, dec);
        if constexpr (DEC16) {
            Dec16Regs d16;
            make_dec16(dec, a.planes_absmax, lane, d16);
            store_dec16_lds(d16, wfwd, lane);
        } else {
            store_decoder_lds(dec, wfwd, lane);
        }
    }
    if constexpr (DEC16) {
        build_grad16_lds(a, w1t, w0t, lane, wave, NWB);
        __syncthreads();
    } else {
        const float g0 = a.decoder_lr_mul * 0.17677669529663687f, g1 = a.decoder_lr_mul * 0.125f;
        for (int i = threadIdx.x; i < 4 * 8 * 64; i += NTHB) {
            const int l = i & 63, st = (i >> 6) & 7, mt = i >> 9, jj = l & 15, gg = l >> 4;
            w1t[i] = a.dec_w1[(1 + 16 * (st >> 2) + 4 * gg + (st & 3)) * 64 + 16 * mt + jj] * g1;
        }
        for (int i = threadIdx.x; i < 2 * 16 * 64; i += NTHB) {
            const int l = i & 63, st = (i >> 6) & 15, ft = i >> 10, jj = l & 15, gg = l >> 4;
            w0t[i] = a.dec_w0[(16 * (st >> 2) + 4 * gg + (st & 3)) * 32 + 16 * ft + jj] * g0;
        }
        __syncthreads();
    }

    GradLds& gl = reinterpret_cast<GradLds*>(glds_raw)[PG ? wave : 0];      // never dereferenced unless PG
    f32x4 aw1[2][4], aw0[4][2];                      // PG: dW1c tiles [ct][kt], dW0 tiles [kt][ft]
    float s_dO[2][4], s_dp[4][4], s_sw[4][4], s_ds = 0.f;
    if constexpr (PG) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                aw1[x][y] = f32x4{0.f, 0.f, 0.f, 0.f};
                aw0[y][x] = f32x4{0.f, 0.f, 0.f, 0.f};
                s_dO[x][y] = 0.f;
            }
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) { s_dp[x][y] = 0.f; s_sw[x][y] = 0.f; }
    }

    // XCD-local schedule over (ray in column-strip order, tile of the ray): raymarch_common.h ray_schedule
    // (WIDE: the schedule runs over RAYS and a wave walks the tiles of its ray in sequence — ray generation and the ray's dL/dfeat
    // once per ray, the depths a tile's gather hangs on loaded a tile ahead — as raymarch_bwd_df_kernel does)
    const RaySchedule sch = ray_schedule((long long)p.total_rays * (WIDE ? 1 : NT), wave, NWB);
    // PIPE (the decoder-gradient-only variant, one wave per SIMD with 512 registers; round 5): software pipeline over the tiles —
    // the saved depths of tile i+2 and the 24 texel loads of tile i+1 are in flight while tile i runs its decoder and its 64
    // weight-gradient MFMAs.  Without it the wave walked  rec load -> taps -> gather -> decoder -> MFMAs  strictly in sequence,
    // alone on its SIMD: 5.7 us per tile, 1.1 ms per 2 frames for 0.16 ms of matrix work.
    constexpr bool PIPE = !SCATTER && !WIDE;
    struct RecPre { int b, ray, tt; float o3[3], d3[3]; float4 rec; float zq; float4 gf[2]; DfSlots slots; };
    struct GatPre { float w[3][4]; float4 v0[3][4], v1[3][4]; };
    auto issue_rec = [&](long long tile) __attribute__((always_inline)) {
        RecPre r;
        const int tt = __builtin_amdgcn_readfirstlane((int)(tile % NT));
        r.tt = tt;
        int pi, pj;
        ray_of(__builtin_amdgcn_readfirstlane((int)(tile / NT)), a.res, r.b, pi, pj);
        r.ray = __builtin_amdgcn_readfirstlane(r.b * R + pi * a.res + pj);
        ray_setup(a, r.b, pi, pj, r.o3, r.d3);
        r.rec = *reinterpret_cast<const float4*>(p.rec + ((size_t)r.ray * S + 16 * tt + j) * 4);
        r.zq = p.rec[((size_t)r.ray * S + 16 * tt + (lane >> 2)) * 4];
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) r.gf[ot] = *reinterpret_cast<const float4*>(p.g_feat + (size_t)r.ray * 32 + 16 * ot + 4 * g);
        if (ro.dfs) r.slots = load_df_slots(ro, (size_t)r.ray * S + 16 * tt + j);
        return r;
    };
    auto issue_gather = [&](const RecPre& r) __attribute__((always_inline)) {
        GatPre q;
        PlaneTaps tq[3];
        sample_taps(p, r.o3, r.d3, r.zq, tq);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const char* base = reinterpret_cast<const char*>(a.planes + ((size_t)(r.b * 3 + pl) * a.H * a.W) * 32);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned off = ((unsigned)tq[pl].idx[k] * 32u + 8u * (lane & 3)) * 4u;
                q.v0[pl][k] = *reinterpret_cast<const float4*>(base + off);
                q.v1[pl][k] = *reinterpret_cast<const float4*>(base + off + 16);
                q.w[pl][k] = tq[pl].w[k];
            }
        }
        return q;
    };
    auto finish_gather = [&](const GatPre& q, float f[8]) __attribute__((always_inline)) {      // (the arithmetic of gather8)
#pragma unroll
        for (int c = 0; c < 8; ++c) f[c] = 0.f;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const float v[4][8] = {{q.v0[pl][0].x, q.v0[pl][0].y, q.v0[pl][0].z, q.v0[pl][0].w, q.v1[pl][0].x, q.v1[pl][0].y, q.v1[pl][0].z, q.v1[pl][0].w},
                                   {q.v0[pl][1].x, q.v0[pl][1].y, q.v0[pl][1].z, q.v0[pl][1].w, q.v1[pl][1].x, q.v1[pl][1].y, q.v1[pl][1].z, q.v1[pl][1].w},
                                   {q.v0[pl][2].x, q.v0[pl][2].y, q.v0[pl][2].z, q.v0[pl][2].w, q.v1[pl][2].x, q.v1[pl][2].y, q.v1[pl][2].z, q.v1[pl][2].w},
                                   {q.v0[pl][3].x, q.v0[pl][3].y, q.v0[pl][3].z, q.v0[pl][3].w, q.v1[pl][3].x, q.v1[pl][3].y, q.v1[pl][3].z, q.v1[pl][3].w}};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float acc = v[0][c] * q.w[pl][0];
                acc = fmaf(v[1][c], q.w[pl][1], acc);
                acc = fmaf(v[2][c], q.w[pl][2], acc);
                acc = fmaf(v[3][c], q.w[pl][3], acc);
                f[c] += acc;
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) f[c] *= 0.3333333333333333f;
    };
    RecPre r_cur, r_nxt;
    GatPre g_cur;
    const long long last_tile = sch.begin < sch.end ? sch.begin + ((sch.end - 1 - sch.begin) / sch.stride) * sch.stride : sch.begin;
    if constexpr (PIPE) {
        if (sch.begin < sch.end) {
            r_cur = issue_rec(sch.begin);
            g_cur = issue_gather(r_cur);
            r_nxt = issue_rec(min(sch.begin + sch.stride, last_tile));
        }
    }
    float o3w[3] = {0.f, 0.f, 0.f}, d3w[3] = {0.f, 0.f, 0.f}, zqw = 0.f;      // WIDE: what a ray's tiles share
    float4 gfw[2] = {};
    int bw = 0, rayw = 0;
    for (long long it = sch.begin; it < sch.end; it += sch.stride) {
#pragma unroll 1
    for (int tw = 0; tw < (WIDE ? NT : 1); ++tw) {
        const long long tile = WIDE ? it * NT + tw : it;
        float4 rec;
        float f[8];
        float4 gfeat[2];
        PlaneTaps taps[3];
        int b = 0, ray = 0, tt = 0;
        DfSlots slots;
        if constexpr (PIPE) {
            // tile i+1: its depths have landed -> taps -> its 24 texel loads go out; tile i+2: its depths go out; THEN tile i
            // (past the wave's last tile the pipeline re-loads that tile: no branch, nothing is used twice)
            const GatPre g_nxt = issue_gather(r_nxt);
            const RecPre r_nn = issue_rec(min(tile + 2 * sch.stride, last_tile));
            __builtin_amdgcn_sched_barrier(0);
            rec = r_cur.rec; gfeat[0] = r_cur.gf[0]; gfeat[1] = r_cur.gf[1];
            ray = r_cur.ray; tt = r_cur.tt; slots = r_cur.slots;
            finish_gather(g_cur, f);
            const int src = 4 * j + g;
#pragma unroll
            for (int c = 0; c < 8; ++c) f[c] = __shfl(f[c], src);
            r_cur = r_nxt; r_nxt = r_nn; g_cur = g_nxt;
        } else {
        tt = WIDE ? tw : __builtin_amdgcn_readfirstlane((int)(tile % NT));        // wave-uniform -> scalar registers
        if (!WIDE || tw == 0) {
            int pi, pj;
            ray_of(__builtin_amdgcn_readfirstlane((int)(WIDE ? it : tile / NT)), a.res, bw, pi, pj);
            rayw = __builtin_amdgcn_readfirstlane(bw * R + pi * a.res + pj);
            ray_setup(a, bw, pi, pj, o3w, d3w);
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) gfw[ot] = *reinterpret_cast<const float4*>(p.g_feat + (size_t)rayw * 32 + 16 * ot + 4 * g);
            zqw = p.rec[((size_t)rayw * S + 16 * tt + (lane >> 2)) * 4];
        }
        b = bw; ray = rayw;
        const float* o3 = o3w; const float* d3 = d3w;
        const int s = 16 * tt + j;
        rec = *reinterpret_cast<const float4*>(p.rec + ((size_t)ray * S + s) * 4);   // depth, omega, dsigma
        if constexpr (SCATTER) sample_taps(p, o3, d3, rec.x, taps);          // taps of sample j: the scatter below publishes them
        {
            // gather in the quad layout of raymarch_kernel (4 adjacent lanes per texel line), then to the MFMA layout
            PlaneTaps tq[3];
            const float zq_cur = zqw;
            if constexpr (WIDE) zqw = p.rec[((size_t)ray * S + 16 * min(tt + 1, NT - 1) + (lane >> 2)) * 4];
            sample_taps(p, o3, d3, zq_cur, tq);
            gather8(a, b, lane & 3, tq, f);
            const int src = 4 * j + g;
#pragma unroll
            for (int c = 0; c < 8; ++c) f[c] = __shfl(f[c], src);
        }
        gfeat[0] = gfw[0]; gfeat[1] = gfw[1];
        if constexpr (!SCATTER) {
            if (ro.dfs) slots = load_df_slots(ro, (size_t)ray * S + s);
        }
        }
        f32x4 hp[4], h[4], o[2];
        float sigma;
        int ln = lane;
        if constexpr (WIDE) asm volatile("" : "+v"(ln));
        if constexpr (DEC16) decoder_fwd16_lds<true>(wfwd, ln, f, hp, h, sigma, o);
        else decoder_fwd_lds<true>(wfwd, ln, f, hp, h, sigma, o);

        // dL/do (colour logits) in the C layout: lane (j, g), register r of tile ot -> channel 16ot + 4g + r
        //   colour = sigmoid(o) * 1.002 - 0.001,  dL/dcolour = omega * 2 dL/dfeat
        f32x4 dO[2];
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
            const float4 gf = gfeat[ot];
            const float gv[4] = {gf.x, gf.y, gf.z, gf.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sg = sigmoid_f(o[ot][r]);
                dO[ot][r] = rec.y * 2.f * gv[r] * 1.002f * sg * (1.f - sg);
            }
        }
        // dH^T = W1c^T . dO^T + wsig (x) dsigma:  A[i][k] = W1[1 + c(k)][16mt + i],  c(k) = 16ot + 4g + r
        f32x4 dH[4];
        f32x4 dF[2];
        if constexpr (DEC16) decoder_bwd16_lds(wfwd, w1t, w0t, ln, dO, rec.z, hp, dH, dF);
#pragma unroll
        for (int mt = 0; mt < (DEC16 ? 0 : 4); ++mt) {
            const float* ws_ = wfwd + (48 + mt * 4) * 64 + ln;
            dH[mt] = f32x4{ws_[0] * rec.z, ws_[64] * rec.z, ws_[128] * rec.z, ws_[192] * rec.z};
#pragma unroll
            for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float wA = w1t[(mt * 8 + ot * 4 + r) * 64 + ln];
                    dH[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA, dO[ot][r], dH[mt], 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) dH[mt][r] *= sigmoid_f(hp[mt][r]);      // softplus' = sigmoid
        }
        // dF^T = W0^T . dHpre^T:  A[i][k] = W0[16mt + 4g + r][16ft + i]
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            if constexpr (!DEC16) {
                dF[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float wA = w0t[(ft * 16 + mt * 4 + r) * 64 + ln];
                        dF[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA, dH[mt][r], dF[ft], 0, 0, 0);
                    }
            }
            // lane (j, g), register r -> feature channel 16ft + 4g + r of sample j
            if constexpr (SCATTER)
                *reinterpret_cast<float4*>(&lds.df[j * 32 + 16 * ft + 4 * g]) =
                    make_float4(dF[ft][0], dF[ft][1], dF[ft][2], dF[ft][3]);
        }
        if constexpr (!SCATTER) {           // sort + gather form of d planes (raymarch_rows.hip): dL/dF to the sample's slots
            if (ro.dfs) store_df_sorted(ro, slots, g, dF);
        }
        if constexpr (PG) {
            // operand images for the weight-gradient products + running bias / sigma-row sums
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gl.sp[(16 * mt + 4 * g + r) * 17 + j] = h[mt][r];
                    gl.dp[(16 * mt + 4 * g + r) * 17 + j] = dH[mt][r];
                    s_dp[mt][r] += dH[mt][r];
                    s_sw[mt][r] += rec.z * h[mt][r];
                }
#pragma unroll
            for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gl.dO[(16 * ot + 4 * g + r) * 17 + j] = dO[ot][r];
                    s_dO[ot][r] += dO[ot][r];
                }
#pragma unroll
            for (int t = 0; t < 8; ++t) gl.f[j * 33 + 8 * g + t] = f[t];
            if (g == 0) s_ds += rec.z;
            WAVE_SYNC();
            // MFMA operands: A[i][k] -> lane (i = j, k = g); B[k][n] -> lane (k = g, n = j); 4 samples per step
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int smp = 4 * st + g;
                float a1[2], bsp[4], a0[4], bf[2];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) a1[ct] = gl.dO[(16 * ct + j) * 17 + smp];
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    bsp[kt] = gl.sp[(16 * kt + j) * 17 + smp];
                    a0[kt] = gl.dp[(16 * kt + j) * 17 + smp];
                }
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) bf[ft] = gl.f[smp * 33 + 16 * ft + j];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
                        aw1[ct][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[ct], bsp[kt], aw1[ct][kt], 0, 0, 0);
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int ft = 0; ft < 2; ++ft)
                        aw0[kt][ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[kt], bf[ft], aw0[kt][ft], 0, 0, 0);
            }
        }
        if constexpr (SCATTER) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)                 // lane (j, g = pl) publishes plane pl's taps of sample j
            if (g == pl) {                             // tap-major [plane*4 + tap][sample]: the scatter reads 16 samples as b128
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    lds.idx[(pl * 4 + k) * 16 + j] = taps[pl].idx[k];
                    lds.wgt[(pl * 4 + k) * 16 + j] = taps[pl].w[k] * 0.3333333333333333f;
                }
            }
        WAVE_SYNC();
        // ---- scatter: lane = channel, half-wave hf walks tap (2kp + hf) of one plane through the 16 depth-sorted
        // samples and MERGES RUNS: consecutive samples whose tap lands on the same texel (the plane the ray is steep
        // to moves < 1 texel per sample; the importance samples cluster) are summed in a register and leave as one
        // 128-byte atomic: 96 -> ~67 atomics per tile on the bench workload (plane (x,y): 3.0 taps per run, the two
        // depth planes: 1.1).  Measured 9.3 -> 8.8 ms per 4 frames; without any atomics the call takes 2.7 ms, i.e.
        // the cost follows the number of DISTINCT lines touched more than the number of atomic instructions.
        {
            const int c = lane & 31, hf = lane >> 5;
            float* base = d_planes + (size_t)b * 3 * a.H * a.W * 32 + c;
            float dfc[16];                             // dL/dfeature[sample][c] of the tile: read once, used by every tap
#pragma unroll
            for (int sm = 0; sm < 16; ++sm) dfc[sm] = lds.df[sm * 32 + c];
#pragma unroll 1
            for (int pk = 0; pk < (MIRROR ? 4 : 6); ++pk) {
                const int pl = pk >> 1, k = 2 * (pk & 1) + hf;
                float* pbase = base + (size_t)pl * a.H * a.W * 32;
                float wv[16];
                int tv[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {          // the 16 samples of this tap: 4 + 4 ds_read_b128 (half-wave broadcast)
                    const float4 w4 = *reinterpret_cast<const float4*>(&lds.wgt[(pl * 4 + k) * 16 + 4 * q]);
                    const int4 t4 = *reinterpret_cast<const int4*>(&lds.idx[(pl * 4 + k) * 16 + 4 * q]);
                    wv[4 * q] = w4.x; wv[4 * q + 1] = w4.y; wv[4 * q + 2] = w4.z; wv[4 * q + 3] = w4.w;
                    tv[4 * q] = t4.x; tv[4 * q + 1] = t4.y; tv[4 * q + 2] = t4.z; tv[4 * q + 3] = t4.w;
                }
                int cur = -1;
                float run = 0.f;
#pragma unroll
                for (int sm = 0; sm < 16; ++sm) {
                    const float wgt = wv[sm];
                    const int t = tv[sm];
                    const float v = dfc[sm] * wgt;
                    if (wgt != 0.f) {
                        if (t == cur) {
                            run += v;
                        } else {
#ifndef HFAGP_NO_ATOMICS
                            if (cur >= 0) unsafeAtomicAdd(pbase + (size_t)cur * 32, run);
#endif
                            cur = t;
                            run = v;
                        }
                    }
                }
#ifndef HFAGP_NO_ATOMICS
                if (cur >= 0) unsafeAtomicAdd(pbase + (size_t)cur * 32, run);
#endif
            }
        }
        }   // SCATTER
        WAVE_SYNC();
    }
    }
    if constexpr (PG) {
        // effective weight = parameter * gain  ->  d parameter = d effective * gain
        const float g0 = a.decoder_lr_mul * 0.17677669529663687f, g1 = a.decoder_lr_mul * 0.125f, gb = a.decoder_lr_mul;
        // C layout: lane (col = j, rows 4g + r)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    unsafeAtomicAdd(dg.w1 + (1 + 16 * ct + 4 * g + r) * 64 + 16 * kt + j, aw1[ct][kt][r] * g1);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int ft = 0; ft < 2; ++ft)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    unsafeAtomicAdd(dg.w0 + (16 * kt + 4 * g + r) * 32 + 16 * ft + j, aw0[kt][ft][r] * g0);
        // per-lane running sums: reduce over the 16 sample lanes (j), lane j == 0 commits
        auto red16 = [](float v) {
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
            return v;
        };
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float vb = red16(s_dp[mt][r]), vw = red16(s_sw[mt][r]);
                if (j == 0) {
                    unsafeAtomicAdd(dg.b0 + 16 * mt + 4 * g + r, vb * gb);
                    unsafeAtomicAdd(dg.w1 + 16 * mt + 4 * g + r, vw * g1);          // sigma row of W1
                }
            }
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = red16(s_dO[ot][r]);
                if (j == 0) unsafeAtomicAdd(dg.b1 + 1 + 16 * ot + 4 * g + r, v * gb);
            }
        const float vs = red16(s_ds);
        if (lane == 0) unsafeAtomicAdd(dg.b1, vs * gb);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Column variant of pass 2 (generator frozen, mirrored planes, H = W <= 256): the scatter into plane (x,z) goes through
// an LDS line cache OWNED by the workgroup instead of global atomics.
//
// Why a column: rays of one image column share their x direction (exactly for a camera without pitch / roll, to a
// few texels over the whole column otherwise), so ALL samples of a 32-ray column chunk project into ONE thin slanted
// band of the (x,z) plane — 32 rays x 96 samples x 4 taps = 12 288 updates onto ~600 texel lines.  Per z row the band
// is the bilinear pair (x0, x0 + 1) of the ray, which moves by ~1/30 texel from one ray of the column to the next, so a
// direct-mapped cache with slot = row * 2 + (col & 1) holds it whatever its slant; when x0 steps over a texel the
// evicted line leaves as one global atomic.  128 B x 512 slots = 64 KB of LDS.
//
// Ownership instead of atomics (LDS float atomics run at 0.33 lanes / clk / CU, a plain read-add-write at 7.3,
// tests/micro/lds_atomic_rate.hip): a round = the 16-sample tiles of kColWaves / NT rays (two rays of six tiles), one
// tile per wave; every wave computes dL/dF of its tile into LDS and publishes, per sample, the two z rows it touches
// (row, x0, the two column weights); after a barrier wave w walks the row-updates with row % kColWaves == w — its rows,
// its cache lines, nobody else's — with half-wave 0 on column x0 and half-wave 1 on x0 + 1 (the two slots of the row).
// Four updates with distinct rows are in flight at a time (their tag / line / dF reads are independent); equal rows
// (the importance samples cluster) fall back to a chain that merges runs in registers.  Plane (x,y) keeps the run-merged
// global atomics (a ray's (x,y) footprint is a line segment no other ray of the column shares).  At the end of the
// chunk every wave flushes its rows with one atomic per valid line.
// Measured (B = 2, 128^2 rays x 96 samples): global atomics cost 1.60 ms in the tile kernel, 0.32 ms here.
constexpr int kColWaves = 12, kColRays = 32, kColSlots = 2, kColMaxRows = 256;

struct ColTileLds {
    float df[16 * 32];                                     // dL/dfeature [sample][channel]
    __attribute__((aligned(16))) int   idx0[4 * 16];       // plane (x,y): texel index [tap][sample]
    __attribute__((aligned(16))) float wgt0[4 * 16];       //              weight / 3
    __attribute__((aligned(16))) int4  upd[2 * 16];        // plane (x,z): [zrow][sample] = (row | -1, x0, bits(w0), bits(w1))
};

template <int S, bool DEC16>
__global__ void __launch_bounds__(kColWaves * 64, 1)
raymarch_bwd_cols_kernel(const RayParams p, float* __restrict__ d_planes, const int nchunks, const int chunks_per_col) {
    constexpr int NT = S / 16, RPR = kColWaves / NT;       // tiles per ray, rays per round
    static_assert(RPR >= 1, "at least one ray per round");
    extern __shared__ __attribute__((aligned(16))) unsigned char cols_smem[];
    float* cache = reinterpret_cast<float*>(cols_smem);                                   // [rows*2][32]
    int* tag = reinterpret_cast<int*>(cache + kColMaxRows * kColSlots * 32);              // [rows*2]: column or -1
    float* w1t = reinterpret_cast<float*>(tag + kColMaxRows * kColSlots);                 // as in raymarch_bwd_tiles_kernel
    float* w0t = w1t + 4 * 8 * 64;
    float* wfwd = w0t + 2 * 16 * 64;
    ColTileLds* tiles = reinterpret_cast<ColTileLds*>(wfwd + kDecLdsRows * 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    ColTileLds& lds = tiles[wave];
    const HfagpRaymarchArgs& a = p.a;
    const int j = lane & 15, g = lane >> 4;
    const int R = a.res * a.res;
    const int nslots = a.H * kColSlots;

    if (wave == 0) {
        DecoderRegs dec;
        load_decoder(a, j, g, dec);
        if constexpr (DEC16) {
            Dec16Regs d16;
            make_dec16(dec, a.planes_absmax, lane, d16);
            store_dec16_lds(d16, wfwd, lane);
        } else {
            store_decoder_lds(dec, wfwd, lane);
        }
    }
    if constexpr (DEC16) {
        build_grad16_lds(a, w1t, w0t, lane, wave, kColWaves);
    } else {
        const float g0 = a.decoder_lr_mul * 0.17677669529663687f, g1 = a.decoder_lr_mul * 0.125f;
        for (int i = threadIdx.x; i < 4 * 8 * 64; i += kColWaves * 64) {
            const int l = i & 63, st = (i >> 6) & 7, mt = i >> 9, jj = l & 15, gg = l >> 4;
            w1t[i] = a.dec_w1[(1 + 16 * (st >> 2) + 4 * gg + (st & 3)) * 64 + 16 * mt + jj] * g1;
        }
        for (int i = threadIdx.x; i < 2 * 16 * 64; i += kColWaves * 64) {
            const int l = i & 63, st = (i >> 6) & 15, ft = i >> 10, jj = l & 15, gg = l >> 4;
            w0t[i] = a.dec_w0[(16 * (st >> 2) + 4 * gg + (st & 3)) * 32 + 16 * ft + jj] * g0;
        }
    }
    const int c = lane & 31, hf = lane >> 5;
    for (int i = threadIdx.x; i < nslots; i += kColWaves * 64) tag[i] = -1;
    __syncthreads();

    // chunk -> (frame, column, first row); consecutive chunks (the pieces of one column, then the next column) run on
    // the same XCD so that the band stays in that XCD's L2
    for (int chunk = (int)xcd_remap(blockIdx.x, gridDim.x); chunk < nchunks; chunk += gridDim.x) {
        const int col_id = chunk / chunks_per_col, piece = chunk % chunks_per_col;
        const int b = col_id / a.res, pj = col_id % a.res;
        const int row0 = piece * kColRays, row1 = min(a.res, row0 + kColRays);
        float* const pb0 = d_planes + (size_t)b * 3 * a.H * a.W * 32 + c;            // plane (x,y), this lane's channel
        float* const pb1 = pb0 + (size_t)a.H * a.W * 32;                             // plane (x,z)
        for (int pr = row0; pr < row1; pr += RPR) {
            const int pi = pr + wave / NT, tt = wave % NT;
            const bool active = wave < RPR * NT && pi < row1;
            if (active) {
                const int ray = __builtin_amdgcn_readfirstlane(b * R + pi * a.res + pj);
                float o3[3], d3[3];
                ray_setup(a, b, pi, pj, o3, d3);
                const int s = 16 * tt + j;
                const float4 rec = *reinterpret_cast<const float4*>(p.rec + ((size_t)ray * S + s) * 4);   // depth, omega, dsigma
                PlaneTaps taps[3];
                sample_taps(p, o3, d3, rec.x, taps);
                float f[8];
                {
                    PlaneTaps tq[3];
                    sample_taps(p, o3, d3, p.rec[((size_t)ray * S + 16 * tt + (lane >> 2)) * 4], tq);
                    gather8(a, b, lane & 3, tq, f);
                    const int src = 4 * j + g;
#pragma unroll
                    for (int cc = 0; cc < 8; ++cc) f[cc] = __shfl(f[cc], src);
                }
                f32x4 hp[4], h[4], o[2];
                float sigma;
                if constexpr (DEC16) decoder_fwd16_lds<true>(wfwd, lane, f, hp, h, sigma, o);
                else decoder_fwd_lds<true>(wfwd, lane, f, hp, h, sigma, o);
                f32x4 dO[2];
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) {
                    const float4 gf = *reinterpret_cast<const float4*>(p.g_feat + (size_t)ray * 32 + 16 * ot + 4 * g);
                    const float gv[4] = {gf.x, gf.y, gf.z, gf.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sg = sigmoid_f(o[ot][r]);
                        dO[ot][r] = rec.y * 2.f * gv[r] * 1.002f * sg * (1.f - sg);
                    }
                }
                f32x4 dH[4];
                f32x4 dF2[2];
                if constexpr (DEC16) decoder_bwd16_lds(wfwd, w1t, w0t, lane, dO, rec.z, hp, dH, dF2);
#pragma unroll
                for (int mt = 0; mt < (DEC16 ? 0 : 4); ++mt) {
                    const float* ws_ = wfwd + (48 + mt * 4) * 64 + lane;
                    dH[mt] = f32x4{ws_[0] * rec.z, ws_[64] * rec.z, ws_[128] * rec.z, ws_[192] * rec.z};
#pragma unroll
                    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float wA = w1t[(mt * 8 + ot * 4 + r) * 64 + lane];
                            dH[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA, dO[ot][r], dH[mt], 0, 0, 0);
                        }
#pragma unroll
                    for (int r = 0; r < 4; ++r) dH[mt][r] *= sigmoid_f(hp[mt][r]);
                }
#pragma unroll
                for (int ft = 0; ft < 2; ++ft) {
                    f32x4 dF = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (DEC16) {
                        dF = dF2[ft];
                    } else {
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float wA = w0t[(ft * 16 + mt * 4 + r) * 64 + lane];
                                dF = __builtin_amdgcn_mfma_f32_16x16x4f32(wA, dH[mt][r], dF, 0, 0, 0);
                            }
                    }
                    *reinterpret_cast<float4*>(&lds.df[j * 32 + 16 * ft + 4 * g]) = make_float4(dF[0], dF[1], dF[2], dF[3]);
                }
                if (g == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        lds.idx0[k * 16 + j] = taps[0].idx[k];
                        lds.wgt0[k * 16 + j] = taps[0].w[k] * 0.3333333333333333f;
                    }
                }
                if (g == 1) {
                    // the two z rows of the sample in plane (x,z): taps (0,1) = row y0, cols x0, x0+1; (2,3) = row y0+1.
                    // row / x0 from a tap that is inside the plane (a clamped index carries weight 0)
#pragma unroll
                    for (int zr = 0; zr < 2; ++zr) {
                        const float wa = taps[1].w[2 * zr] * 0.3333333333333333f, wb = taps[1].w[2 * zr + 1] * 0.3333333333333333f;
                        int row = -1, x0 = 0;
                        if (wa != 0.f) { row = taps[1].idx[2 * zr] / a.W; x0 = taps[1].idx[2 * zr] % a.W; }
                        else if (wb != 0.f) { row = taps[1].idx[2 * zr + 1] / a.W; x0 = taps[1].idx[2 * zr + 1] % a.W - 1; }
                        lds.upd[zr * 16 + j] = make_int4(row, x0, __float_as_int(wa), __float_as_int(wb));
                    }
                }
                WAVE_SYNC();
                // ---- plane (x,y): run-merged global atomics, as in raymarch_bwd_tiles_kernel
                {
                    float dfc[16];
#pragma unroll
                    for (int sm = 0; sm < 16; ++sm) dfc[sm] = lds.df[sm * 32 + c];
#pragma unroll 1
                    for (int pk = 0; pk < 2; ++pk) {
                        const int k = 2 * pk + hf;
                        float wv[16];
                        int tv[16];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 w4 = *reinterpret_cast<const float4*>(&lds.wgt0[k * 16 + 4 * q]);
                            const int4 t4 = *reinterpret_cast<const int4*>(&lds.idx0[k * 16 + 4 * q]);
                            wv[4 * q] = w4.x; wv[4 * q + 1] = w4.y; wv[4 * q + 2] = w4.z; wv[4 * q + 3] = w4.w;
                            tv[4 * q] = t4.x; tv[4 * q + 1] = t4.y; tv[4 * q + 2] = t4.z; tv[4 * q + 3] = t4.w;
                        }
                        int cur = -1;
                        float run = 0.f;
#pragma unroll
                        for (int sm = 0; sm < 16; ++sm) {
                            const float wgt = wv[sm];
                            const int t = tv[sm];
                            const float v = dfc[sm] * wgt;
                            if (wgt != 0.f) {
                                if (t == cur) {
                                    run += v;
                                } else {
#ifndef HFAGP_NO_ATOMICS
                                    if (cur >= 0) unsafeAtomicAdd(pb0 + (size_t)cur * 32, run);
#endif
                                    cur = t;
                                    run = v;
                                }
                            }
                        }
#ifndef HFAGP_NO_ATOMICS
                        if (cur >= 0) unsafeAtomicAdd(pb0 + (size_t)cur * 32, run);
#endif
                    }
                }
            } else if (lane < 32) {
                lds.upd[lane] = make_int4(-1, 0, 0, 0);      // an idle wave of the round publishes no row-updates
            }
            __syncthreads();                        // dF and the row-updates of the round's tiles are published
            // ---- plane (x,z): wave w applies the row-updates of ITS rows to its cache lines
            {
                constexpr int NU = kColWaves * 32;   // row-updates of the round: [wave][zrow][sample]
                // one update of this half-wave's column: tag check (a different column in the slot leaves as ONE atomic),
                // accumulate; `have` / `accv` are the slot's tag and line as read from LDS
                auto apply = [&](int row, int col, float wgt, float dfv, int have, float accv) {
                    const int slot = row * kColSlots + (col & 1);
                    if (have != col) {
#ifndef HFAGP_NO_ATOMICS
                        if (have >= 0) unsafeAtomicAdd(pb1 + ((size_t)row * a.W + have) * 32, accv);
#endif
                        accv = 0.f;
                        if (c == 0) tag[slot] = col;
                    }
                    cache[slot * 32 + c] = fmaf(dfv, wgt, accv);
                };
#pragma unroll 1
                for (int e0 = 0; e0 < NU; e0 += 64) {
                    const int u = e0 + lane;
                    const int4 me = tiles[u >> 5].upd[u & 31];
                    const bool mine = me.x >= 0 && (me.x % kColWaves) == wave;
                    unsigned long long mask = __ballot(mine);
                    while (mask) {
                        int ii[4], rw[4], n = 0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            ii[k] = 0; rw[k] = -1 - k;
                            if (mask) {
                                ii[k] = __builtin_ctzll(mask);
                                mask &= mask - 1;
                                rw[k] = __builtin_amdgcn_readlane(me.x, ii[k]);
                                n = k + 1;
                            }
                        }
                        const bool distinct = rw[0] != rw[1] && rw[0] != rw[2] && rw[0] != rw[3] && rw[1] != rw[2] &&
                                              rw[1] != rw[3] && rw[2] != rw[3];
                        int colk[4], havek[4];
                        float wk[4], dk[4], acck[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int x0 = __builtin_amdgcn_readlane(me.y, ii[k]);
                            const float wa = __int_as_float(__builtin_amdgcn_readlane(me.z, ii[k]));
                            const float wb = __int_as_float(__builtin_amdgcn_readlane(me.w, ii[k]));
                            const int ue = e0 + ii[k];
                            colk[k] = x0 + hf;
                            wk[k] = k < n ? (hf ? wb : wa) : 0.f;
                            dk[k] = tiles[ue >> 5].df[(ue & 15) * 32 + c];
                        }
                        if (distinct) {
                            // four different rows = eight different slots: all reads first, then the updates
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int slot = max(rw[k], 0) * kColSlots + (colk[k] & 1);
                                havek[k] = tag[slot];
                                acck[k] = cache[slot * 32 + c];
                            }
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (wk[k] != 0.f) apply(rw[k], colk[k], wk[k], dk[k], havek[k], acck[k]);
                        } else {
                            // equal rows in the batch (clustered samples): one at a time, in order
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (wk[k] != 0.f) {
                                    const int slot = rw[k] * kColSlots + (colk[k] & 1);
                                    apply(rw[k], colk[k], wk[k], dk[k], tag[slot], cache[slot * 32 + c]);
                                }
                        }
                    }
                }
            }
            __syncthreads();                        // the next round overwrites dF / the row-updates
        }
        // ---- end of the chunk: every wave flushes its rows (one atomic per valid line) and invalidates them
        for (int row = wave; row < a.H; row += kColWaves) {
            const int slot = row * kColSlots + hf;
            const int col = tag[slot];
#ifndef HFAGP_NO_ATOMICS
            if (col >= 0) unsafeAtomicAdd(pb1 + ((size_t)row * a.W + col) * 32, cache[slot * 32 + c]);
#endif
        }
        WAVE_SYNC();
        for (int row = wave; row < a.H; row += kColWaves)
            if (lane < kColSlots) tag[row * kColSlots + lane] = -1;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 4: pass 2 of the column variant as TWO kernels.  In raymarch_bwd_cols_kernel the 12 waves of the one workgroup a CU
// holds (64 KB cache + 44 KB decoder images + tiles) move in lock step — tile arithmetic | barrier | scatter | barrier — so
// the gather / MFMA latencies of the arithmetic and the LDS / atomic latencies of the two scatters never overlap: 0.6 ms of
// arithmetic + 0.95 ms of scatter per 2 frames (profiles/r03_raybwd_phases.txt), each alone a third of that.  Split:
//   raymarch_bwd_df_kernel      gather -> decoder forward -> decoder backward per 16-sample tile, dL/dF to a scratch buffer
//                               [ray][sample][32] (201 MB per frame); no barrier in its loop, two 4-wave workgroups per CU;
//   raymarch_bwd_scatter_kernel the column chunk's scatter alone: tiles stream dL/dF back in (2 KB per tile, one pass), taps are
//                               recomputed from the saved depths, plane (x,y) run-merged atomics + plane (x,z) line cache as before.
// The scratch round trip is 0.4 GB per frame of streaming traffic; taken when the caller provides HfagpRaymarchBwdArgs::df_scratch.
constexpr int kDfWaves = 4;      // two workgroups per CU (254 VGPRs: two waves per SIMD, as the forward kernel)

template <int S, bool DEC16, bool SORTED>
__global__ void __launch_bounds__(kDfWaves * 64, SORTED ? 3 : 2)
raymarch_bwd_df_kernel(const RayParams p, float* __restrict__ df_out, const RowsOut ro) {
    __shared__ float w1t[4 * 8 * 64];
    __shared__ float w0t[2 * 16 * 64];
    __shared__ float wfwd[kDecLdsRows * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const HfagpRaymarchArgs& a = p.a;
    const int j = lane & 15, g = lane >> 4;
    const int R = a.res * a.res;
    constexpr int NT = S / 16;
    if (wave == 0) {
        DecoderRegs dec;
        load_decoder(a, j, g, dec);
        if constexpr (DEC16) {
            Dec16Regs d16;
            make_dec16(dec, a.planes_absmax, lane, d16);
            store_dec16_lds(d16, wfwd, lane);
        } else {
            store_decoder_lds(dec, wfwd, lane);
        }
    }
    if constexpr (DEC16) {
        build_grad16_lds(a, w1t, w0t, lane, wave, kDfWaves);
    } else {
        const float g0 = a.decoder_lr_mul * 0.17677669529663687f, g1 = a.decoder_lr_mul * 0.125f;
        for (int i = threadIdx.x; i < 4 * 8 * 64; i += kDfWaves * 64) {
            const int l = i & 63, st = (i >> 6) & 7, mt = i >> 9, jj = l & 15, gg = l >> 4;
            w1t[i] = a.dec_w1[(1 + 16 * (st >> 2) + 4 * gg + (st & 3)) * 64 + 16 * mt + jj] * g1;
        }
        for (int i = threadIdx.x; i < 2 * 16 * 64; i += kDfWaves * 64) {
            const int l = i & 63, st = (i >> 6) & 15, ft = i >> 10, jj = l & 15, gg = l >> 4;
            w0t[i] = a.dec_w0[(16 * (st >> 2) + 4 * gg + (st & 3)) * 32 + 16 * ft + jj] * g0;
        }
    }
    __syncthreads();
    // a wave walks the tiles of ONE ray in sequence: ray generation (six divisions and a square root), the position -> pixel
    // arithmetic and the ray's dL/dfeat once per ray instead of once per tile (~100 of ~800 vector instructions of a tile)
    const RaySchedule sch = ray_schedule((long long)p.total_rays, wave, kDfWaves);
    for (long long rp = sch.begin; rp < sch.end; rp += sch.stride) {
        int b, pi, pj;
        ray_of(__builtin_amdgcn_readfirstlane((int)rp), a.res, b, pi, pj);
        const int ray = __builtin_amdgcn_readfirstlane(b * R + pi * a.res + pj);
        float o3[3], d3[3];
        ray_setup(a, b, pi, pj, o3, d3);
        float4 gfeat[2];
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) gfeat[ot] = *reinterpret_cast<const float4*>(p.g_feat + (size_t)ray * 32 + 16 * ot + 4 * g);
        // the depth the gather of a tile hangs on is loaded a tile ahead (the chain  depth -> taps -> 24 texel loads -> decoder
        // is what a wave waits on: two waves per SIMD hide one of its two memory round trips, not both)
        float zq = p.rec[((size_t)ray * S + (lane >> 2)) * 4];
#pragma unroll 1
        for (int tt = 0; tt < NT; ++tt) {
        const int s = 16 * tt + j;
        int ln = lane;
        asm volatile("" : "+v"(ln));       // the weight images are READ per tile (not hoisted into ~170 registers): 3 waves per SIMD
        const float zq_cur = zq;
        zq = p.rec[((size_t)ray * S + 16 * min(tt + 1, NT - 1) + (lane >> 2)) * 4];
        const float4 rec = *reinterpret_cast<const float4*>(p.rec + ((size_t)ray * S + s) * 4);   // depth, omega, dsigma
        DfSlots slots;
        if constexpr (SORTED) slots = load_df_slots(ro, (size_t)ray * S + s);      // lands while the tile runs its decoder
        float f[8];
        {
            PlaneTaps tq[3];
            sample_taps(p, o3, d3, zq_cur, tq);
            gather8(a, b, lane & 3, tq, f);
            const int src = 4 * j + g;
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) f[cc] = __shfl(f[cc], src);
        }
        f32x4 hp[4], h[4], o[2];
        float sigma;
        if constexpr (DEC16) decoder_fwd16_lds<true>(wfwd, ln, f, hp, h, sigma, o);
        else decoder_fwd_lds<true>(wfwd, lane, f, hp, h, sigma, o);
        f32x4 dO[2];
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
            const float4 gf = gfeat[ot];
            const float gv[4] = {gf.x, gf.y, gf.z, gf.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sg = sigmoid_f(o[ot][r]);
                dO[ot][r] = rec.y * 2.f * gv[r] * 1.002f * sg * (1.f - sg);
            }
        }
        f32x4 dH[4];
        f32x4 dF2[2];
        if constexpr (DEC16) decoder_bwd16_lds(wfwd, w1t, w0t, ln, dO, rec.z, hp, dH, dF2);
#pragma unroll
        for (int mt = 0; mt < (DEC16 ? 0 : 4); ++mt) {
            const float* ws_ = wfwd + (48 + mt * 4) * 64 + lane;
            dH[mt] = f32x4{ws_[0] * rec.z, ws_[64] * rec.z, ws_[128] * rec.z, ws_[192] * rec.z};
#pragma unroll
            for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float wA = w1t[(mt * 8 + ot * 4 + r) * 64 + lane];
                    dH[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA, dO[ot][r], dH[mt], 0, 0, 0);
                }
#pragma unroll
            for (int r = 0; r < 4; ++r) dH[mt][r] *= sigmoid_f(hp[mt][r]);
        }
        // lane (j, g), register r -> feature channel 16ft + 4g + r of sample j: 16-byte stores, two per 128-byte sample line
        float* dst = df_out + ((size_t)ray * S + s) * 32 + 4 * g;
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            if constexpr (!DEC16) {
                dF2[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float wA = w0t[(ft * 16 + mt * 4 + r) * 64 + lane];
                        dF2[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA, dH[mt][r], dF2[ft], 0, 0, 0);
                    }
            }
            if constexpr (!SORTED) *reinterpret_cast<float4*>(dst + 16 * ft) = make_float4(dF2[ft][0], dF2[ft][1], dF2[ft][2], dF2[ft][3]);
        }
        if constexpr (SORTED) store_df_sorted(ro, slots, g, dF2);                     // sort + gather form: raymarch_rows.hip
        }
    }
}

template <int S>
__global__ void __launch_bounds__(kColWaves * 64, 1)
raymarch_bwd_scatter_kernel(const RayParams p, const float* __restrict__ df_in, float* __restrict__ d_planes, const int nchunks,
                            const int chunks_per_col) {
    constexpr int NT = S / 16, RPR = kColWaves / NT;       // tiles per ray, rays per round
    static_assert(RPR >= 1, "at least one ray per round");
    extern __shared__ __attribute__((aligned(16))) unsigned char cols_smem[];
    float* cache = reinterpret_cast<float*>(cols_smem);                                   // [rows*2][32]
    int* tag = reinterpret_cast<int*>(cache + kColMaxRows * kColSlots * 32);              // [rows*2]: column or -1
    ColTileLds* tiles = reinterpret_cast<ColTileLds*>(tag + kColMaxRows * kColSlots);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    ColTileLds& lds = tiles[wave];
    const HfagpRaymarchArgs& a = p.a;
    const int j = lane & 15, g = lane >> 4;
    const int R = a.res * a.res;
    const int nslots = a.H * kColSlots;
    const int c = lane & 31, hf = lane >> 5;
    for (int i = threadIdx.x; i < nslots; i += kColWaves * 64) tag[i] = -1;
    __syncthreads();

    for (int chunk = (int)xcd_remap(blockIdx.x, gridDim.x); chunk < nchunks; chunk += gridDim.x) {
        const int col_id = chunk / chunks_per_col, piece = chunk % chunks_per_col;
        const int b = col_id / a.res, pj = col_id % a.res;
        const int row0 = piece * kColRays, row1 = min(a.res, row0 + kColRays);
        float* const pb0 = d_planes + (size_t)b * 3 * a.H * a.W * 32 + c;            // plane (x,y), this lane's channel
        float* const pb1 = pb0 + (size_t)a.H * a.W * 32;                             // plane (x,z)
        for (int pr = row0; pr < row1; pr += RPR) {
            const int pi = pr + wave / NT, tt = wave % NT;
            const bool active = wave < RPR * NT && pi < row1;
            if (active) {
                const int ray = __builtin_amdgcn_readfirstlane(b * R + pi * a.res + pj);
                float o3[3], d3[3];
                ray_setup(a, b, pi, pj, o3, d3);
                const int s = 16 * tt + j;
                // the tile's dL/dF: 2 KB, contiguous in the scratch buffer -> LDS [sample][channel] as is
                {
                    const float4* src = reinterpret_cast<const float4*>(df_in + ((size_t)ray * S + 16 * tt) * 32);
                    const float4 v0 = src[lane], v1 = src[lane + 64];
                    reinterpret_cast<float4*>(lds.df)[lane] = v0;
                    reinterpret_cast<float4*>(lds.df)[lane + 64] = v1;
                }
                const float depth = p.rec[((size_t)ray * S + s) * 4];
                PlaneTaps taps[3];
                sample_taps(p, o3, d3, depth, taps);
                if (g == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        lds.idx0[k * 16 + j] = taps[0].idx[k];
                        lds.wgt0[k * 16 + j] = taps[0].w[k] * 0.3333333333333333f;
                    }
                }
                if (g == 1) {
#pragma unroll
                    for (int zr = 0; zr < 2; ++zr) {
                        const float wa = taps[1].w[2 * zr] * 0.3333333333333333f, wb = taps[1].w[2 * zr + 1] * 0.3333333333333333f;
                        int row = -1, x0 = 0;
                        if (wa != 0.f) { row = taps[1].idx[2 * zr] / a.W; x0 = taps[1].idx[2 * zr] % a.W; }
                        else if (wb != 0.f) { row = taps[1].idx[2 * zr + 1] / a.W; x0 = taps[1].idx[2 * zr + 1] % a.W - 1; }
                        lds.upd[zr * 16 + j] = make_int4(row, x0, __float_as_int(wa), __float_as_int(wb));
                    }
                }
                WAVE_SYNC();
                // ---- plane (x,y): run-merged global atomics
                {
                    float dfc[16];
#pragma unroll
                    for (int sm = 0; sm < 16; ++sm) dfc[sm] = lds.df[sm * 32 + c];
#pragma unroll 1
                    for (int pk = 0; pk < 2; ++pk) {
                        const int k = 2 * pk + hf;
                        float wv[16];
                        int tv[16];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 w4 = *reinterpret_cast<const float4*>(&lds.wgt0[k * 16 + 4 * q]);
                            const int4 t4 = *reinterpret_cast<const int4*>(&lds.idx0[k * 16 + 4 * q]);
                            wv[4 * q] = w4.x; wv[4 * q + 1] = w4.y; wv[4 * q + 2] = w4.z; wv[4 * q + 3] = w4.w;
                            tv[4 * q] = t4.x; tv[4 * q + 1] = t4.y; tv[4 * q + 2] = t4.z; tv[4 * q + 3] = t4.w;
                        }
                        int cur = -1;
                        float run = 0.f;
#pragma unroll
                        for (int sm = 0; sm < 16; ++sm) {
                            const float wgt = wv[sm];
                            const int t = tv[sm];
                            const float v = dfc[sm] * wgt;
                            if (wgt != 0.f) {
                                if (t == cur) {
                                    run += v;
                                } else {
                                    if (cur >= 0) unsafeAtomicAdd(pb0 + (size_t)cur * 32, run);
                                    cur = t;
                                    run = v;
                                }
                            }
                        }
                        if (cur >= 0) unsafeAtomicAdd(pb0 + (size_t)cur * 32, run);
                    }
                }
            } else if (lane < 32) {
                lds.upd[lane] = make_int4(-1, 0, 0, 0);
            }
            __syncthreads();
            // ---- plane (x,z): wave w applies the row-updates of ITS rows to its cache lines (as raymarch_bwd_cols_kernel)
            {
                constexpr int NU = kColWaves * 32;
                auto apply = [&](int row, int col, float wgt, float dfv, int have, float accv) {
                    const int slot = row * kColSlots + (col & 1);
                    if (have != col) {
                        if (have >= 0) unsafeAtomicAdd(pb1 + ((size_t)row * a.W + have) * 32, accv);
                        accv = 0.f;
                        if (c == 0) tag[slot] = col;
                    }
                    cache[slot * 32 + c] = fmaf(dfv, wgt, accv);
                };
#pragma unroll 1
                for (int e0 = 0; e0 < NU; e0 += 64) {
                    const int u = e0 + lane;
                    const int4 me = tiles[u >> 5].upd[u & 31];
                    const bool mine = me.x >= 0 && (me.x % kColWaves) == wave;
                    unsigned long long mask = __ballot(mine);
                    while (mask) {
                        int ii[4], rw[4], n = 0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            ii[k] = 0; rw[k] = -1 - k;
                            if (mask) {
                                ii[k] = __builtin_ctzll(mask);
                                mask &= mask - 1;
                                rw[k] = __builtin_amdgcn_readlane(me.x, ii[k]);
                                n = k + 1;
                            }
                        }
                        const bool distinct = rw[0] != rw[1] && rw[0] != rw[2] && rw[0] != rw[3] && rw[1] != rw[2] &&
                                              rw[1] != rw[3] && rw[2] != rw[3];
                        int colk[4], havek[4];
                        float wk[4], dk[4], acck[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int x0 = __builtin_amdgcn_readlane(me.y, ii[k]);
                            const float wa = __int_as_float(__builtin_amdgcn_readlane(me.z, ii[k]));
                            const float wb = __int_as_float(__builtin_amdgcn_readlane(me.w, ii[k]));
                            const int ue = e0 + ii[k];
                            colk[k] = x0 + hf;
                            wk[k] = k < n ? (hf ? wb : wa) : 0.f;
                            dk[k] = tiles[ue >> 5].df[(ue & 15) * 32 + c];
                        }
                        if (distinct) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int slot = max(rw[k], 0) * kColSlots + (colk[k] & 1);
                                havek[k] = tag[slot];
                                acck[k] = cache[slot * 32 + c];
                            }
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (wk[k] != 0.f) apply(rw[k], colk[k], wk[k], dk[k], havek[k], acck[k]);
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (wk[k] != 0.f) {
                                    const int slot = rw[k] * kColSlots + (colk[k] & 1);
                                    apply(rw[k], colk[k], wk[k], dk[k], tag[slot], cache[slot * 32 + c]);
                                }
                        }
                    }
                }
            }
            __syncthreads();
        }
        for (int row = wave; row < a.H; row += kColWaves) {
            const int slot = row * kColSlots + hf;
            const int col = tag[slot];
            if (col >= 0) unsafeAtomicAdd(pb1 + ((size_t)row * a.W + col) * 32, cache[slot * 32 + c]);
        }
        WAVE_SYNC();
        for (int row = wave; row < a.H; row += kColWaves)
            if (lane < kColSlots) tag[row * kColSlots + lane] = -1;
        __syncthreads();
    }
}

// d_planes[b][2][x][z][:] = d_planes[b][1][z][x][:]   (one float4 per thread, full 128-byte lines both ways)
__global__ void __launch_bounds__(256) mirror_plane_kernel(float* __restrict__ d_planes, int B, int N) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long long)B * N * N * 8) return;
    const int c4 = (int)(tid & 7);
    const long long pix = tid >> 3;
    const int z = (int)(pix % N), x = (int)((pix / N) % N), b = (int)(pix / ((long long)N * N));
    const float4* src = reinterpret_cast<const float4*>(d_planes + ((((size_t)b * 3 + 1) * N + z) * N + x) * 32);
    float4* dst = reinterpret_cast<float4*>(d_planes + ((((size_t)b * 3 + 2) * N + x) * N + z) * 32);
    dst[c4] = src[c4];
}

template <int S, bool DEC16>
static void launch_tiles2(bool pg, bool mirror, unsigned blocks, const RayParams& p, float* d_planes, const DecGrads& dg,
                          hipStream_t s) {
    if (pg) {
        if (mirror) raymarch_bwd_tiles_kernel<S, true, true, DEC16><<<blocks, BwdWaves<true>::value * 64, 0, s>>>(p, d_planes, dg, RowsOut{});
        else raymarch_bwd_tiles_kernel<S, true, false, DEC16><<<blocks, BwdWaves<true>::value * 64, 0, s>>>(p, d_planes, dg, RowsOut{});
    } else {
        if (mirror) raymarch_bwd_tiles_kernel<S, false, true, DEC16><<<blocks, BwdWaves<false>::value * 64, 0, s>>>(p, d_planes, dg, RowsOut{});
        else raymarch_bwd_tiles_kernel<S, false, false, DEC16><<<blocks, BwdWaves<false>::value * 64, 0, s>>>(p, d_planes, dg, RowsOut{});
    }
}

// decoder-parameter gradients only (d planes is the column kernel's job)
template <int S>
static void launch_decoder_grads(unsigned blocks, const RayParams& p, float* d_planes, const DecGrads& dg, const RowsOut& ro,
                                 hipStream_t s) {
    // two waves per SIMD reading the weight images from LDS (WIDE): 1.06 -> 0.93 ms per 2 frames against one 512-register wave
    // per SIMD with the software pipeline
    static const bool narrow = getenv("HFAGP_DEV_PG_NARROW") != nullptr;      // developer switch: A/B timing
    if (!narrow) {
        if (p.a.planes_absmax)
            raymarch_bwd_tiles_kernel<S, true, true, true, false, true><<<blocks, 512, 0, s>>>(p, d_planes, dg, ro);
        else
            raymarch_bwd_tiles_kernel<S, true, true, false, false, true><<<blocks, 512, 0, s>>>(p, d_planes, dg, ro);
        return;
    }
    if (p.a.planes_absmax)
        raymarch_bwd_tiles_kernel<S, true, true, true, false><<<blocks, TileWaves<true, false>::value * 64, 0, s>>>(p, d_planes, dg, ro);
    else
        raymarch_bwd_tiles_kernel<S, true, true, false, false><<<blocks, TileWaves<true, false>::value * 64, 0, s>>>(p, d_planes, dg, ro);
}

// sort + gather form: dL/dF of every sample to its slots (generator frozen: no decoder gradients)
template <int S>
static void launch_df_sorted(const RayParams& p, const RowsOut& ro, hipStream_t s) {
    const long long ntiles = (long long)p.total_rays * (S / 16);
    static const int per_cu = getenv("HFAGP_DEV_DF_BLOCKS") ? atoi(getenv("HFAGP_DEV_DF_BLOCKS")) : 6;      // developer: A/B timing
    const unsigned dblocks = (unsigned)std::min<long long>((ntiles + kDfWaves - 1) / kDfWaves, (long long)kNumCU * per_cu);
    if (p.a.planes_absmax) raymarch_bwd_df_kernel<S, true, true><<<dblocks, kDfWaves * 64, 0, s>>>(p, nullptr, ro);
    else raymarch_bwd_df_kernel<S, false, true><<<dblocks, kDfWaves * 64, 0, s>>>(p, nullptr, ro);
}

template <int S>
static void launch_tiles(bool pg, bool mirror, unsigned blocks, const RayParams& p, float* d_planes, const DecGrads& dg,
                         hipStream_t s) {
    if (p.a.planes_absmax) launch_tiles2<S, true>(pg, mirror, blocks, p, d_planes, dg, s);
    else launch_tiles2<S, false>(pg, mirror, blocks, p, d_planes, dg, s);
}

static size_t scatter_lds_bytes() {
    return (size_t)kColMaxRows * kColSlots * 32 * sizeof(float) + (size_t)kColMaxRows * kColSlots * sizeof(int) +
           kColWaves * sizeof(ColTileLds);
}

static size_t cols_lds_bytes() {
    return (size_t)kColMaxRows * kColSlots * 32 * sizeof(float) + (size_t)kColMaxRows * kColSlots * sizeof(int) +
           (size_t)(4 * 8 * 64 + 2 * 16 * 64 + kDecLdsRows * 64) * sizeof(float) + kColWaves * sizeof(ColTileLds);
}

template <int S, bool DEC16>
static int launch_cols2(unsigned blocks, size_t lds, const RayParams& p, float* d_planes, int nchunks, int chunks_per_col,
                        hipStream_t s) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&raymarch_bwd_cols_kernel<S, DEC16>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        set_error("raymarch_bwd: cannot raise dynamic LDS to %zu bytes: %s", lds, hipGetErrorString(e));
        return HFAGP_ELAUNCH;
    }
    raymarch_bwd_cols_kernel<S, DEC16><<<blocks, kColWaves * 64, lds, s>>>(p, d_planes, nchunks, chunks_per_col);
    return HFAGP_OK;
}

// the two-kernel form of the column variant (df_scratch given): dL/dF to the scratch buffer, then the scatter alone
template <int S>
static int launch_df_scatter(const RayParams& p, float* df, float* d_planes, unsigned cblocks, int nchunks, int chunks_per_col,
                             hipStream_t s) {
    const long long ntiles = (long long)p.total_rays * (S / 16);
    const unsigned dblocks = (unsigned)std::min<long long>((ntiles + kDfWaves - 1) / kDfWaves, (long long)kNumCU * 2 * 8);
    if (p.a.planes_absmax) raymarch_bwd_df_kernel<S, true, false><<<dblocks, kDfWaves * 64, 0, s>>>(p, df, RowsOut{});
    else raymarch_bwd_df_kernel<S, false, false><<<dblocks, kDfWaves * 64, 0, s>>>(p, df, RowsOut{});
    const size_t lds = scatter_lds_bytes();
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&raymarch_bwd_scatter_kernel<S>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        set_error("raymarch_bwd: cannot raise dynamic LDS to %zu bytes: %s", lds, hipGetErrorString(e));
        return HFAGP_ELAUNCH;
    }
    raymarch_bwd_scatter_kernel<S><<<cblocks, kColWaves * 64, lds, s>>>(p, df, d_planes, nchunks, chunks_per_col);
    return HFAGP_OK;
}

template <int S>
static int launch_cols(unsigned blocks, size_t lds, const RayParams& p, float* d_planes, int nchunks, int chunks_per_col,
                       hipStream_t s) {
    return p.a.planes_absmax ? launch_cols2<S, true>(blocks, lds, p, d_planes, nchunks, chunks_per_col, s)
                             : launch_cols2<S, false>(blocks, lds, p, d_planes, nchunks, chunks_per_col, s);
}

}  // namespace hfagp

using namespace hfagp;

extern "C" int hfagp_raymarch_bwd(const HfagpRaymarchBwdArgs* a, void* stream) {
    HFAGP_REQUIRE(a && a->g_feat && a->d_planes && a->rec, HFAGP_EBADARG, "raymarch_bwd: null pointer");
    RayParams p;
    int rc = fill_ray_params(&a->fwd, p, "raymarch_bwd");
    if (rc != HFAGP_OK) return rc;
    p.g_feat = a->g_feat;
    p.rec = a->rec;
    hipStream_t s = (hipStream_t)stream;
    rc = launch_raymarch(p, true, s);
    if (rc != HFAGP_OK) return rc;
    const int S = a->fwd.Sc + a->fwd.Sf;
    const long long ntiles = (long long)p.total_rays * (S / 16);
    DecGrads dg{a->d_dec_w0, a->d_dec_b0, a->d_dec_w1, a->d_dec_b1};
    const bool pg = a->d_dec_w0 != nullptr;
    const int nwb = pg ? BwdWaves<true>::value : BwdWaves<false>::value;
    long long blocks = (ntiles + nwb - 1) / nwb;
    // (decoder gradients: one resident workgroup per CU and ~4.3 k end-of-kernel atomics per wave on the same buffers —
    // the grid is the chip; the ray schedule strides over the rest)
    const long long cap = pg ? (long long)kNumCU : (long long)kNumCU * 2 * 8;
    if (blocks > cap) blocks = cap;
    HFAGP_REQUIRE(!pg || (a->d_dec_b0 && a->d_dec_w1 && a->d_dec_b1), HFAGP_EBADARG,
                  "raymarch_bwd: decoder gradients need all four buffers");
    // planes 1 and 2 mirror each other for EG3D's original axes on square planes: scatter plane 1 only
    const bool mirror = a->fwd.plane_axes == 0 && a->fwd.H == a->fwd.W;
    // sort + gather form (raymarch_rows.hip): the caller provided its scratch buffer
    if (a->rows_scratch) {
        RowsOut ro;
        rc = rows_prepare(p, a->rows_scratch, a->rows_scratch_bytes, ro, s);
        if (rc != HFAGP_OK) return rc;
        if (pg) {
            // decoder MLP gradients + dL/dF in one pass (one wave per SIMD: the pass that was the decoder gradients alone)
            const unsigned gblocks = (unsigned)std::min<long long>((ntiles + 3) / 4, (long long)kNumCU);
            if (S == 96) launch_decoder_grads<96>(gblocks, p, a->d_planes, dg, ro, s);
            else if (S == 64) launch_decoder_grads<64>(gblocks, p, a->d_planes, dg, ro, s);
            else launch_decoder_grads<32>(gblocks, p, a->d_planes, dg, ro, s);
        } else if (S == 96) launch_df_sorted<96>(p, ro, s);
        else if (S == 64) launch_df_sorted<64>(p, ro, s);
        else launch_df_sorted<32>(p, ro, s);
        rc = rows_gather(p, a->rows_scratch, a->d_planes, s);
        if (rc != HFAGP_OK) return rc;
        if (mirror) {
            const long long n = (long long)a->fwd.B * a->fwd.H * a->fwd.W * 8;
            mirror_plane_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(a->d_planes, a->fwd.B, a->fwd.H);
        }
        return check_launch("raymarch_bwd/rows");
    }
    // frozen generator + mirrored square planes up to 256^2: the column variant (LDS line cache for plane (x,z))
    static const bool no_cols = getenv("HFAGP_DEV_NO_COLS") != nullptr;      // developer switch: A/B timing
    if (mirror && a->fwd.H <= kColMaxRows && !no_cols) {
        const int chunks_per_col = (a->fwd.res + kColRays - 1) / kColRays;
        const int nchunks = a->fwd.B * a->fwd.res * chunks_per_col;
        const size_t lds = cols_lds_bytes();
        unsigned cblocks = (unsigned)std::min<long long>(nchunks, (long long)kNumCU * 4);
        int rcl = HFAGP_OK;
        if (a->df_scratch) {
            if (S == 96) rcl = launch_df_scatter<96>(p, a->df_scratch, a->d_planes, cblocks, nchunks, chunks_per_col, s);
            else if (S == 64) rcl = launch_df_scatter<64>(p, a->df_scratch, a->d_planes, cblocks, nchunks, chunks_per_col, s);
            else rcl = launch_df_scatter<32>(p, a->df_scratch, a->d_planes, cblocks, nchunks, chunks_per_col, s);
        }
        else if (S == 96) rcl = launch_cols<96>(cblocks, lds, p, a->d_planes, nchunks, chunks_per_col, s);
        else if (S == 64) rcl = launch_cols<64>(cblocks, lds, p, a->d_planes, nchunks, chunks_per_col, s);
        else rcl = launch_cols<32>(cblocks, lds, p, a->d_planes, nchunks, chunks_per_col, s);
        if (rcl != HFAGP_OK) return rcl;
        if (pg) {                   // generator being tuned: the decoder-parameter gradients in a pass of their own
            // one resident workgroup per CU (109 KB of LDS); every wave ends with ~4.3 k atomics on the SAME gradient
            // buffers, so the grid is the chip, not a multiple of it (x16: 1.90 ms, x1: 1.09 ms per B = 2 call)
            const unsigned gblocks = (unsigned)std::min<long long>((ntiles + 3) / 4, (long long)kNumCU);
            if (S == 96) launch_decoder_grads<96>(gblocks, p, a->d_planes, dg, RowsOut{}, s);
            else if (S == 64) launch_decoder_grads<64>(gblocks, p, a->d_planes, dg, RowsOut{}, s);
            else launch_decoder_grads<32>(gblocks, p, a->d_planes, dg, RowsOut{}, s);
        }
    } else if (S == 96) launch_tiles<96>(pg, mirror, (unsigned)blocks, p, a->d_planes, dg, s);
    else if (S == 64) launch_tiles<64>(pg, mirror, (unsigned)blocks, p, a->d_planes, dg, s);
    else launch_tiles<32>(pg, mirror, (unsigned)blocks, p, a->d_planes, dg, s);
    if (mirror) {
        const long long n = (long long)a->fwd.B * a->fwd.H * a->fwd.W * 8;
        mirror_plane_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(a->d_planes, a->fwd.B, a->fwd.H);
    }
    return check_launch("raymarch_bwd/tiles");
}

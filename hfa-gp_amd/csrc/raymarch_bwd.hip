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
//   The importance depths carry no gradient (EG3D: no_grad + detach), neither do the camera / depths.
#include "raymarch_common.h"

namespace hfagp {

struct TileLds {
    float df[16 * 32];        // dL/dfeature of the 16 samples, [sample][channel]
    int   idx[16 * 12];       // texel index of every tap, [sample][plane*4 + tap]
    float wgt[16 * 12];       // bilinear weight / 3
};

template <int S>
__global__ void __launch_bounds__(256, 2) raymarch_bwd_tiles_kernel(const RayParams p, float* __restrict__ d_planes) {
    __shared__ TileLds lds_all[4];
    // A operands of the two backward products, lane-linear ([step][lane]: conflict-free ds_read_b32), shared by
    // the 4 waves:  w1t[mt][ot*4+r][lane] = W1[1 + 16ot + 4g + r][16mt + j] * g1
    //               w0t[ft][mt*4+r][lane] = W0[16mt + 4g + r][16ft + j] * g0
    __shared__ float w1t[4 * 8 * 64];
    __shared__ float w0t[2 * 16 * 64];
    __shared__ float wfwd[kDecLdsRows * 64];       // forward A operands (DecoderRegs image)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    TileLds& lds = lds_all[wave];
    const HfagpRaymarchArgs& a = p.a;
    const int j = lane & 15, g = lane >> 4;
    const int R = a.res * a.res;
    constexpr int NT = S / 16;

    if (wave == 0) {
        DecoderRegs dec;
        load_decoder(a, j, g, dec);
        store_decoder_lds(dec, wfwd, lane);
    }
    {
        const float g0 = a.decoder_lr_mul * 0.17677669529663687f, g1 = a.decoder_lr_mul * 0.125f;
        for (int i = threadIdx.x; i < 4 * 8 * 64; i += 256) {
            const int l = i & 63, st = (i >> 6) & 7, mt = i >> 9, jj = l & 15, gg = l >> 4;
            w1t[i] = a.dec_w1[(1 + 16 * (st >> 2) + 4 * gg + (st & 3)) * 64 + 16 * mt + jj] * g1;
        }
        for (int i = threadIdx.x; i < 2 * 16 * 64; i += 256) {
            const int l = i & 63, st = (i >> 6) & 15, ft = i >> 10, jj = l & 15, gg = l >> 4;
            w0t[i] = a.dec_w0[(16 * (st >> 2) + 4 * gg + (st & 3)) * 32 + 16 * ft + jj] * g0;
        }
        __syncthreads();
    }

    const long long ntiles = (long long)p.total_rays * NT;
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long long)gridDim.x * 4) {
        const int ray = (int)(tile / NT), tt = (int)(tile % NT);
        const int b = ray / R, rr = ray % R;
        float o3[3], d3[3];
        ray_setup(a, b, rr / a.res, rr % a.res, o3, d3);
        const int s = 16 * tt + j;
        const float4 rec = *reinterpret_cast<const float4*>(p.rec + ((size_t)ray * S + s) * 4);   // depth, omega, dsigma
        PlaneTaps taps[3];
        sample_taps(p, o3, d3, rec.x, taps);
        float f[8];
        gather8(a, b, g, taps, f);
        f32x4 hp[4], h[4], o[2];
        float sigma;
        decoder_fwd_lds<true>(wfwd, lane, f, hp, h, sigma, o);

        // dL/do (colour logits) in the C layout: lane (j, g), register r of tile ot -> channel 16ot + 4g + r
        //   colour = sigmoid(o) * 1.002 - 0.001,  dL/dcolour = omega * 2 dL/dfeat
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
        // dH^T = W1c^T . dO^T + wsig (x) dsigma:  A[i][k] = W1[1 + c(k)][16mt + i],  c(k) = 16ot + 4g + r
        f32x4 dH[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
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
            for (int r = 0; r < 4; ++r) dH[mt][r] *= sigmoid_f(hp[mt][r]);      // softplus' = sigmoid
        }
        // dF^T = W0^T . dHpre^T:  A[i][k] = W0[16mt + 4g + r][16ft + i]
        f32x4 dF[2];
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
            dF[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float wA = w0t[(ft * 16 + mt * 4 + r) * 64 + lane];
                    dF[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA, dH[mt][r], dF[ft], 0, 0, 0);
                }
            // lane (j, g), register r -> feature channel 16ft + 4g + r of sample j
            *reinterpret_cast<float4*>(&lds.df[j * 32 + 16 * ft + 4 * g]) =
                make_float4(dF[ft][0], dF[ft][1], dF[ft][2], dF[ft][3]);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)                 // lane (j, g = pl) publishes plane pl's taps of sample j
            if (g == pl) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    lds.idx[j * 12 + pl * 4 + k] = taps[pl].idx[k];
                    lds.wgt[j * 12 + pl * 4 + k] = taps[pl].w[k] * 0.3333333333333333f;
                }
            }
        WAVE_SYNC();
        // ---- scatter: half-wave hf handles samples hf, hf+2, ...; lane = channel
        {
            const int c = lane & 31, hf = lane >> 5;
            float* base = d_planes + (size_t)b * 3 * a.H * a.W * 32 + c;
            for (int sm = hf; sm < 16; sm += 2) {
                const float v = lds.df[sm * 32 + c];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float wgt = lds.wgt[sm * 12 + pl * 4 + k];
#ifndef HFAGP_NO_ATOMICS
                        if (wgt != 0.f)
#else
                        if (wgt == 12345.f)
#endif
                            unsafeAtomicAdd(base + ((size_t)pl * a.H * a.W + lds.idx[sm * 12 + pl * 4 + k]) * 32, v * wgt);
                    }
            }
        }
        WAVE_SYNC();
    }
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
    long long blocks = (ntiles + 3) / 4;
    const long long cap = (long long)kNumCU * 2 * 8;
    if (blocks > cap) blocks = cap;
    if (S == 96) raymarch_bwd_tiles_kernel<96><<<(unsigned)blocks, 256, 0, s>>>(p, a->d_planes);
    else if (S == 64) raymarch_bwd_tiles_kernel<64><<<(unsigned)blocks, 256, 0, s>>>(p, a->d_planes);
    else raymarch_bwd_tiles_kernel<32><<<(unsigned)blocks, 256, 0, s>>>(p, a->d_planes);
    return check_launch("raymarch_bwd/tiles");
}

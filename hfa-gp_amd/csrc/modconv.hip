// Modulated convolution as an implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32), gfx950.
//
//   M = output pixels of one 8x16 (or 4x16 ...) spatial patch, N = output channels, K = taps x Cin.
//   x is channels-last; the patch (+halo) for an 8-channel K-chunk is staged in LDS once and
//   re-read for every tap (9x reuse); the style modulation is applied on the way into LDS and the
//   demodulation / noise / bias / leaky-ReLU / clamp in the epilogue, so the weights are shared by
//   the whole batch (EG3D's "scale activations" form of modulated_conv2d).
//   The up-sampling conv runs as the 4 output phases of the stride-2 transposed convolution
//   (4+2+2+1 = 9 taps in total, i.e. no zero-insert FLOPs); its FIR is hfagp_upfir_epilogue_fwd.
//
//   K order inside a chunk is permuted so that each lane fetches its A and B operands for four
//   consecutive MFMAs with one ds_read_b128 each:  MFMA step s uses channel 4*h + s for the lane
//   half h = lane>>5  (A[i][k]: lane = 32*k + i,  B[k][j]: lane = 32*k + j).
#include "modconv_plan.h"

namespace hfagp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CK = 8;            // input channels per K-chunk
constexpr int AS = 12;           // LDS pixel stride of the A patch in floats (8 + pad, 16-B aligned)
constexpr int LPW = 32;          // LDS row pitch of the patch in pixels: with AS = 12 (3 x 16-B slots, odd) every
                                 // 16-lane group of a ds_read_b128 then hits 16 distinct slots (no bank conflicts)

template <int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(256) modconv_kernel(const ConvParams p) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32, PH = BM / PW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* As = lds;                                   // [ph*pw][AS]
    float* Bs = lds + ((PH + 2) * LPW) * AS;           // [ntaps][2][BN][4]

    const Phase& ph = p.phase[blockIdx.y];
    // ---- decode the block id: N tile fastest (neighbours share the input patch in L2)
    // (readfirstlane: run-time divisions are vector ops; keep what derives from the block coordinates in SGPRs)
    unsigned id = blockIdx.x;
    const int tn_blk = __builtin_amdgcn_readfirstlane(id % p.tiles_n); id /= p.tiles_n;
    const int tw = __builtin_amdgcn_readfirstlane(id % p.tiles_w);     id /= p.tiles_w;
    const int th = __builtin_amdgcn_readfirstlane(id % p.tiles_h);     id /= p.tiles_h;
    const int b = __builtin_amdgcn_readfirstlane(id % p.B);            id /= p.B;
    const int ks = __builtin_amdgcn_readfirstlane(id);
    const int m0 = th * PH, n0 = tw * PW, co0 = tn_blk * BN;
    if (m0 >= ph.mh || n0 >= ph.mw) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5, l31 = lane & 31;

    const int c_begin = __builtin_amdgcn_readfirstlane((int)(((long long)p.nchunks * ks) / p.ksplit));
    const int c_end = __builtin_amdgcn_readfirstlane((int)(((long long)p.nchunks * (ks + 1)) / p.ksplit));

    // ---- staging assignment (register prefetch: global -> regs -> LDS)
    const int npatch = p.ph * p.pw;
    constexpr int A_PER_T = ((PH + 2) * (PW + 2) * 2 + 255) / 256;   // float4 per thread, A
    constexpr int B_PER_T = (MAXTAPS * 2 * BN + 255) / 256;          // float4 per thread, B
    float4 ra[A_PER_T], rs[A_PER_T], rb[B_PER_T];
    const float* xb = p.x + ph.in_off + (long long)b * p.x_batch_stride;
    const float* sb = p.styles ? p.styles + (size_t)b * p.Cin : nullptr;
    const int nB = ph.ntaps * 2 * BN;
    const int cq = p.Cin >> 2;

    // chunk-independent source offsets (-1 = zero fill)
    long long aoff[A_PER_T], boff[B_PER_T];
    int lds_a[A_PER_T];          // LDS word offset of each staged float4 of the patch
#pragma unroll
    for (int k = 0; k < A_PER_T; ++k) {
        const int idx = tid + k * 256;
        aoff[k] = -1;
        lds_a[k] = 0;
        if (idx < npatch * 2) {
            const int pix = idx >> 1, q = idx & 1;
            lds_a[k] = ((pix / p.pw) * LPW + pix % p.pw) * AS + 4 * q;
            const int iy = m0 + p.dymin + pix / p.pw, ix = n0 + p.dxmin + pix % p.pw;
            if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) aoff[k] = ((long long)iy * p.in_w + ix) * p.Cin + 4 * q;
        }
    }
#pragma unroll
    for (int k = 0; k < B_PER_T; ++k) {
        const int idx = tid + k * 256;
        boff[k] = -1;
        if (idx < nB) {
            const int co = idx % BN, q = (idx / BN) & 1, t = idx / (2 * BN);
            if (co0 + co < p.Cout) boff[k] = ((long long)ph.widx[t] * cq + q) * p.Cout + co0 + co;
        }
    }

    auto load_regs = [&](int chunk) {
        const int c0 = chunk * CK;
#pragma unroll
        for (int k = 0; k < A_PER_T; ++k) {
            // no use of the loaded values here: the style multiply happens in store_lds, so the loads of the
            // chunk stay in flight under the MFMAs (a use right after a load costs an exposed L2 round trip)
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f), sv = make_float4(1.f, 1.f, 1.f, 1.f);
            if (aoff[k] >= 0) {
                v = *reinterpret_cast<const float4*>(xb + aoff[k] + c0);
                if (sb) sv = *reinterpret_cast<const float4*>(sb + c0 + 4 * ((tid + k * 256) & 1));
            }
            ra[k] = v; rs[k] = sv;
        }
#pragma unroll
        for (int k = 0; k < B_PER_T; ++k) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (boff[k] >= 0) v = reinterpret_cast<const float4*>(p.wt)[boff[k] + (long long)(c0 >> 2) * p.Cout];
            rb[k] = v;
        }
    };
    auto store_lds = [&]() {
#pragma unroll
        for (int k = 0; k < A_PER_T; ++k) {
            const int idx = tid + k * 256;
            if (idx < npatch * 2)
                *reinterpret_cast<float4*>(As + lds_a[k]) =
                    make_float4(ra[k].x * rs[k].x, ra[k].y * rs[k].y, ra[k].z * rs[k].z, ra[k].w * rs[k].w);
        }
#pragma unroll
        for (int k = 0; k < B_PER_T; ++k) {
            const int idx = tid + k * 256;
            if (idx < nB) reinterpret_cast<float4*>(Bs)[idx] = rb[k];
        }
    };

    // ---- per-lane fragment addresses
    int apix[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int pidx = (wm * TM + tm) * 32 + l31;
        apix[tm] = ((pidx >> 4) * LPW + (pidx & 15)) * AS + 4 * h;
    }
    int bcol[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bcol[tn] = (h * BN + (wn * TN + tn) * 32 + l31) * 4;

    // tap table -> registers once (reading it from the kernel argument inside the K loop costs a memory round
    // trip per tap per chunk)
    int toffs[MAXTAPS];
#pragma unroll
    for (int t = 0; t < MAXTAPS; ++t)
        toffs[t] = t < ph.ntaps ? ((ph.dy[t] - p.dymin) * LPW + (ph.dx[t] - p.dxmin)) * AS : 0;
    const int ntaps = ph.ntaps;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    if (c_begin < c_end) load_regs(c_begin);
    for (int c = c_begin; c < c_end; ++c) {
        store_lds();
        __syncthreads();
        if (c + 1 < c_end) load_regs(c + 1);
#pragma unroll
        for (int t = 0; t < MAXTAPS; ++t) {
            if (t >= ntaps) break;
            const int toff = toffs[t];
            float4 a4[TM], b4[TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) a4[tm] = *reinterpret_cast<const float4*>(As + apix[tm] + toff);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b4[tn] = *reinterpret_cast<const float4*>(Bs + t * 2 * BN * 4 + bcol[tn]);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[tm].x, b4[tn].x, acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[tm].y, b4[tn].y, acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[tm].z, b4[tn].z, acc[tm][tn], 0, 0, 0);
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[tm].w, b4[tn].w, acc[tm][tn], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    // ---- epilogue.  C/D layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* out = p.out + (size_t)(ks * p.nslab + ph.slab) * p.slab;
    float vmax = 0.f;                              // max |y| of this lane's stores (fp16 range tracking, hfagp.h)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int co = co0 + (wn * TN + tn) * 32 + l31;
        if (co >= p.Cout) continue;
        float d = 1.f, bs = 0.f;
        if (p.fused) {
            if (p.dcoef) d = p.dcoef[(size_t)b * p.Cout + co];
            if (p.bias) bs = p.bias[co];
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            float nz[16];                          // noise of the 16 rows first: independent loads, one wait
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pidx = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int m = min(m0 + (pidx >> 4), ph.mh - 1), n = min(n0 + (pidx & 15), ph.mw - 1);
                nz[r] = (p.fused && p.noise) ? p.noise[(size_t)(ph.sy * m + ph.oy0) * p.Wo + ph.sx * n + ph.ox0] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int pidx = (wm * TM + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int m = m0 + (pidx >> 4), n = n0 + (pidx & 15);
                if (m >= ph.mh || n >= ph.mw) continue;
                const int oy = ph.sy * m + ph.oy0, ox = ph.sx * n + ph.ox0;
                float v = acc[tm][tn][r];
                if (p.fused) {
                    v = v * d + bs + nz[r] * p.noise_strength;
                    v = lrelu_gain_clamp(v, p.act, p.alpha, p.gain, p.clamp);
                }
                vmax = fmaxf(vmax, fabsf(v));
                out[(((size_t)b * p.Ho + oy) * p.Wo + ox) * p.Cout + co] = v;
            }
        }
    }
    if (p.fused && p.y_absmax) publish_absmax(p.y_absmax, vmax, blockIdx.x * 4 + (threadIdx.x >> 6));
}

// sum the split-K slabs and (optionally) apply the epilogue
__global__ void __launch_bounds__(256) splitk_epilogue_kernel(const float* __restrict__ ws, float* __restrict__ y,
                                                              const float* __restrict__ dcoef,
                                                              const float* __restrict__ noise,
                                                              const float* __restrict__ bias, long long slab,
                                                              int ksplit, int HoWo, int Cout, int fused, int act,
                                                              float noise_strength, float alpha, float gain,
                                                              float clamp, float* __restrict__ y_absmax) {
    const long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i4 * 4 >= slab) return;
    // the slabs are summed in order (bitwise repeatable), but eight loads are in flight at a time: a plain loop waited for
    // every L2 round trip in turn (14 us per call at ksplit 64, 18 calls per frame at B = 1)
    float4 s = reinterpret_cast<const float4*>(ws)[i4];
    int k = 1;
    for (; k + 8 <= ksplit; k += 8) {
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = reinterpret_cast<const float4*>(ws + (size_t)(k + q) * slab)[i4];
#pragma unroll
        for (int q = 0; q < 8; ++q) { s.x += v[q].x; s.y += v[q].y; s.z += v[q].z; s.w += v[q].w; }
    }
    for (; k < ksplit; ++k) {
        const float4 v = reinterpret_cast<const float4*>(ws + (size_t)k * slab)[i4];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (fused) {
        const long long e = i4 * 4;
        const int co = (int)(e % Cout);
        const long long pix = e / Cout;
        const int b = (int)(pix / HoWo);
        float4 d = make_float4(1.f, 1.f, 1.f, 1.f), bs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (dcoef) d = *reinterpret_cast<const float4*>(dcoef + (size_t)b * Cout + co);
        if (bias) bs = *reinterpret_cast<const float4*>(bias + co);
        const float nz = noise ? noise[pix % HoWo] * noise_strength : 0.f;
        s.x = lrelu_gain_clamp(s.x * d.x + bs.x + nz, act, alpha, gain, clamp);
        s.y = lrelu_gain_clamp(s.y * d.y + bs.y + nz, act, alpha, gain, clamp);
        s.z = lrelu_gain_clamp(s.z * d.z + bs.z + nz, act, alpha, gain, clamp);
        s.w = lrelu_gain_clamp(s.w * d.w + bs.w + nz, act, alpha, gain, clamp);
    }
    reinterpret_cast<float4*>(y)[i4] = s;
    if (fused && y_absmax)
        publish_absmax(y_absmax, fmaxf(fmaxf(fabsf(s.x), fabsf(s.y)), fmaxf(fabsf(s.z), fabsf(s.w))), blockIdx.x * 4 + (threadIdx.x >> 6));
}

}  // namespace hfagp

using namespace hfagp;

extern "C" {

static int ck_of(const HfagpModconvArgs* a) { return a && a->precision != HFAGP_PREC_F32 ? 16 : CK; }

size_t hfagp_modconv_workspace_bytes(const HfagpModconvArgs* a) {
    if (validate(a, ck_of(a)) != HFAGP_OK) return 0;
    if (smallconv_takes(a)) {
        const int ks = smallconv_ksplit(a);
        return ks > 1 ? (size_t)ks * a->B * a->H * a->W * a->Cout * sizeof(float) : 0;
    }
    Plan pl;
    if (make_plan(a, pl, ck_of(a)) != HFAGP_OK) return 0;
    return pl.ws_bytes;
}

int32_t hfagp_modconv_rgb_parts(const HfagpModconvArgs* a) { return a ? ((a->Cout + 127) / 128) * 2 : 0; }

int hfagp_modconv_fwd(const HfagpModconvArgs* a, void* stream) {
    int rc = validate(a, ck_of(a));
    if (rc != HFAGP_OK) return rc;
    Plan pl;
    rc = make_plan(a, pl, ck_of(a));
    if (rc != HFAGP_OK) return rc;
    ConvParams& p = pl.p;
    HFAGP_REQUIRE((a->rgb_w == nullptr) == (a->rgb_part == nullptr), HFAGP_EBADARG, "modconv: rgb_w and rgb_part go together");
    if (smallconv_takes(a)) {
        rc = launch_smallconv(a, pl, (hipStream_t)stream);
        if (rc != HFAGP_OK || p.ksplit == 1) return rc;
        const long long n4 = p.slab / 4;           // the slabs: summed (in order) and finished by the reducer, as below
        splitk_epilogue_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(
            a->workspace, a->y, a->dcoef, a->noise, a->bias, p.slab, p.ksplit, p.Ho * p.Wo, a->Cout, 1, a->act,
            a->noise_strength, a->alpha, a->gain, a->clamp, a->y_absmax);
        return check_launch("modconv_fwd/splitk (small-image kernel)");
    }
    HFAGP_REQUIRE(!a->rgb_part || (a->precision != HFAGP_PREC_F32 && p.ksplit * p.nslab == 1 && p.fused && a->Cout % 128 == 0 &&
                                   (a->mode == HFAGP_CONV3X3 || a->mode == HFAGP_CONV1X1)),
                  HFAGP_EUNSUPPORTED, "modconv: the fused toRGB needs a 16-bit precision, mode 0 / 2, Cout %% 128 == 0 and no "
                                      "split-K (workspace_bytes() == 0); got precision %d mode %d Cout %d ksplit %d",
                  a->precision, a->mode, a->Cout, p.ksplit);
    HFAGP_REQUIRE(a->y || (a->rgb_part && !a->y_absmax), HFAGP_EBADARG,
                  "modconv: y may only be NULL with the fused toRGB (rgb_w / rgb_part) and without y_absmax");
    hipStream_t s = (hipStream_t)stream;
    const int nslabs = p.ksplit * p.nslab;
    if (nslabs > 1) {
        HFAGP_REQUIRE(a->workspace, HFAGP_EBADARG, "modconv: split-K (%d slabs) needs a workspace of %zu bytes", nslabs,
                      pl.ws_bytes);
        p.out = a->workspace;
    } else {
        p.out = a->y;
    }
    HFAGP_REQUIRE(!(a->x_f16 || a->y_f16) || a->precision == HFAGP_PREC_F16, HFAGP_EUNSUPPORTED,
                  "modconv: fp16 storage (x_f16 / y_f16) goes with precision HFAGP_PREC_F16");
    if (a->precision != HFAGP_PREC_F32) {
        rc = launch_modconv_bf16(a, pl, s);
        if (rc != HFAGP_OK) return rc;
    } else {
        const size_t lds_bytes = ((size_t)(pl.bm / PW + 2) * LPW * AS + (size_t)MAXTAPS * 2 * pl.bn * 4) * sizeof(float);
        switch (pl.bn) {
            case 128: modconv_kernel<2, 2, 2, 2><<<pl.grid, 256, lds_bytes, s>>>(p); break;
            case 96:  modconv_kernel<4, 1, 1, 3><<<pl.grid, 256, lds_bytes, s>>>(p); break;
            case 64:  modconv_kernel<2, 2, 2, 1><<<pl.grid, 256, lds_bytes, s>>>(p); break;
            default:  modconv_kernel<4, 1, 1, 1><<<pl.grid, 256, lds_bytes, s>>>(p); break;
        }
        rc = check_launch("modconv_fwd");
        if (rc != HFAGP_OK) return rc;
    }
    if (nslabs > 1) {
        const int fused = a->mode != HFAGP_CONVT3X3_UP2 && a->mode != HFAGP_CONVS2_BWD;
        const long long n4 = p.slab / 4;
        splitk_epilogue_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, s>>>(
            a->workspace, a->y, a->dcoef, a->noise, a->bias, p.slab, nslabs, p.Ho * p.Wo, a->Cout, fused, a->act,
            a->noise_strength, a->alpha, a->gain, a->clamp, a->y_absmax);
        rc = check_launch("modconv_fwd/splitk");
    }
    return rc;
}

}  // extern "C"

// toRGB + skip connection of a backbone block as one streaming kernel (hfagp_torgb_skip_fwd, include/hfagp.h).
//
// BUILD NOTE: this translation unit — since round 3 EVERY unit — is compiled with -fno-slp-vectorize (build.sh).  With the SLP vectoriser on, hipcc
// (ROCm 7.2) packs the epilogue's tap arithmetic into v_pk_fma_f32 with swapped op_sel halves next to v_mov writes of
// the swapped source register, and on the MI355X the low-half result of such an instruction is sporadically not written
// for lanes 48-63 (one upsample tap of ~1e-5 of the outputs missing, different positions every run; found by running the
// kernel 100 times against hfagp_modconv_fwd + hfagp_skip_upsample_add, bit for bit: tests/test_gpu_round2.py).
// Scalar v_fma_f32 code is exact in 800 runs.
#include "split_mfma.h"

namespace hfagp {

// toRGB (1x1 modulated conv, linear, no demodulation) + skip connection of a backbone block as a STREAMING kernel
// (hfagp_torgb_skip_fwd).  A 1x1 conv has no patch to share between positions, so nothing is staged through LDS and there
// is no barrier in the K loop: a wave owns 32 consecutive positions of one image row (M = 32), its lanes read their
// position's 8 channels of the K step (32 B; the two lane halves take the two halves of the 16-channel step) straight
// from HBM into registers, scale by the style, split and feed the MFMAs; the B fragments come from the (L2-resident) split
// weight image exactly as in modconv_bf16_kernel.  The accumulators (TNC x 16 registers, column = channel) then take the
// bias and the four taps of upsample2d(img_in) and are stored as 128-B runs: x is read once, img_out written once, the toRGB
// output itself never exists in memory.  The x loads run two K steps ahead (three float4 pairs in flight per lane) and 12
// waves per CU keep ~70 KB in flight, the depth HBM needs.  Same operand arithmetic and accumulation order as the 1-tap
// instance of modconv_bf16_kernel without split-K: identical bits.
template <int KD, int TNC>
__global__ void __launch_bounds__(256, 3) torgb_skip_kernel(const HfagpTorgbSkipArgs a) {
    constexpr int NP = kind_parts(KD);
    constexpr bool F16 = kind_f16(KD);
    constexpr int NPROD = NP == 1 ? 1 : NP == 2 ? 3 : 6;
    constexpr int PA[6] = {0, 1, 0, 1, 2, 0};
    constexpr int PB[6] = {0, 0, 1, 1, 0, 2};
    __shared__ __attribute__((aligned(16))) float Ss[512];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const int HW = a.H * a.W;
    const long long pos0 = ((long long)blockIdx.x * 4 + wave) * 32;                  // first position of the wave's tile
    const int b = __builtin_amdgcn_readfirstlane((int)(((long long)blockIdx.x * 128) / HW));   // H*W % 128 == 0
    for (int i = tid; i < a.Cin; i += 256) Ss[i] = a.styles[(size_t)b * a.Cin + i];
    float sback = 1.f, sdown = 1.f;
    if constexpr (F16) sdown = style_range_guard(a.styles + (size_t)b * a.Cin, a.Cin, lane, &sback, a.x_absmax);
    __syncthreads();

    const int nchunks = a.Cin / CKB;
    const char* xl = reinterpret_cast<const char*>(a.x + (size_t)(pos0 + l31) * a.Cin + 8 * h);      // this lane's row
    const char* wb = reinterpret_cast<const char*>(a.wt);
    const int cq8 = a.Cin >> 3;
    const long long part_stride = (long long)cq8 * a.Cout;                                            // uint4 per part
    unsigned bth[TNC];
#pragma unroll
    for (int tn = 0; tn < TNC; ++tn) bth[tn] = (unsigned)(h * a.Cout + tn * 32 + l31) * 16u;

    f32x16 acc[TNC];
#pragma unroll
    for (int tn = 0; tn < TNC; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tn][r] = 0.f;

    float4 xr[3][2];                                          // ring: K steps c, c+1, c+2
    auto load_x = [&](int c, int slot) __attribute__((always_inline)) {
        const char* q = xl + (size_t)min(c, nchunks - 1) * (CKB * 4);
        xr[slot][0] = *reinterpret_cast<const float4*>(q);
        xr[slot][1] = *reinterpret_cast<const float4*>(q + 16);
    };
    u32x4 bq[TNC][NP];
    auto load_b = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const char* base = wb + (q * part_stride + (long long)c * 2 * a.Cout) * 16;
#pragma unroll
            for (int tn = 0; tn < TNC; ++tn) bq[tn][q] = *reinterpret_cast<const u32x4*>(base + bth[tn]);
        }
    };
    auto step = [&](int c, int slot) __attribute__((always_inline)) {
        load_b(c);
        const float4 s0 = *reinterpret_cast<const float4*>(Ss + c * CKB + 8 * h);
        const float4 s1 = *reinterpret_cast<const float4*>(Ss + c * CKB + 8 * h + 4);
        const float4 x0 = xr[slot][0], x1 = xr[slot][1];
        uint2 p0[NP], p1[NP];
        split4<KD>(make_float4(x0.x * (s0.x * sdown), x0.y * (s0.y * sdown), x0.z * (s0.z * sdown), x0.w * (s0.w * sdown)), p0);
        split4<KD>(make_float4(x1.x * (s1.x * sdown), x1.y * (s1.y * sdown), x1.z * (s1.z * sdown), x1.w * (s1.w * sdown)), p1);
        load_x(c + 3, slot);                                  // the slot is free again: refill it three steps ahead
        u32x4 af[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) af[q] = u32x4{p0[q].x, p0[q].y, p1[q].x, p1[q].y};
#pragma unroll
        for (int pr = 0; pr < NPROD; ++pr)
#pragma unroll
            for (int tn = 0; tn < TNC; ++tn) acc[tn] = mfma16<F16>(af[PA[pr]], bq[tn][PB[pr]], acc[tn]);
    };
    load_x(0, 0);
    load_x(1, 1);
    load_x(2, 2);
    int c = 0;
    for (; c + 3 <= nchunks; c += 3) {
        step(c, 0);
        step(c + 1, 1);
        step(c + 2, 2);
    }
    if (c < nchunks) step(c, 0);
    if (c + 1 < nchunks) step(c + 1, 1);

    // ---- epilogue: C/D layout of 32x32: column (channel) = lane & 31, row (position) = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // A lane's 16 rows are four runs of four consecutive columns X = 4m .. 4m+3; their upsample taps are the four
    // half-resolution columns 2m-1 .. 2m+2 of two half-resolution rows: 8 unconditional loads per run (clamped addresses,
    // a tap outside the image gets weight 0 — fmaf(0, v, o) = o, the bits of the conditional add of skip_kernel), all
    // issued before the first use so that their latencies overlap.
    const int pix0 = (int)(pos0 - (long long)b * HW);          // W % 32 == 0: the 32 positions share the image row
    const int Y = pix0 / a.W, X0 = pix0 % a.W;
    const int Hi = a.H >> 1, Wi = a.W >> 1;
    int y0, y1; float wy0, wy1;
    up2_taps(Y, y0, y1, wy0, wy1);
    if (y0 < 0) { y0 = 0; wy0 = 0.f; }
    if (y1 >= Hi) { y1 = max(Hi - 1, 0); wy1 = 0.f; }
    float vmax = 0.f;
#pragma unroll
    for (int tn = 0; tn < TNC; ++tn) {
        const int co = tn * 32 + l31;
        const float bs = a.bias[co];
        float* dst = a.plane_major ? a.img_out + ((((size_t)b * 3 + tn) * a.H + Y) * a.W + X0) * 32 + l31
                                   : a.img_out + ((size_t)b * HW + pix0) * a.Cout + co;
        const int pstep = a.plane_major ? 32 : a.Cout;
        float o[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = fmaf(acc[tn][r], sback, bs);
        if (a.img_in) {
            const float* src0 = a.img_in + ((size_t)b * Hi + y0) * Wi * a.Cout + co;
            const float* src1 = a.img_in + ((size_t)b * Hi + y1) * Wi * a.Cout + co;
            float v0[4][4], v1[4][4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cb = (X0 >> 1) + 4 * g + 2 * h;      // half-resolution column of X = X0 + 8g + 4h
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int col = cb - 1 + k;
                    const int off = min(max(col, 0), Wi - 1) * a.Cout;   // (32-bit: one sample's image is < 2^31 elements)
                    const bool in = col >= 0 && col < Wi;
                    const float t0 = src0[off], t1 = src1[off];
                    v0[g][k] = in ? t0 : 0.f;                   // a column outside the image contributes +0
                    v1[g][k] = in ? t1 : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int g = r >> 2, q = r & 3;
                // X = 4m + q: even q: taps (2m + q/2 - 1, 2m + q/2) weights (1/4, 3/4); odd q: (2m + (q-1)/2, +1) weights (3/4, 1/4);
                // in units of k = column - (2m - 1): q=0: k 0,1  q=1: k 1,2  q=2: k 1,2  q=3: k 2,3
                const int k0 = (q + 1) >> 1, k1 = k0 + 1;
                const float wx0 = (q & 1) ? 0.75f : 0.25f, wx1 = (q & 1) ? 0.25f : 0.75f;
                float t = o[r];
                t = fmaf(wy0 * wx0, v0[g][k0], t);
                t = fmaf(wy0 * wx1, v0[g][k1], t);
                t = fmaf(wy1 * wx0, v1[g][k0], t);
                t = fmaf(wy1 * wx1, v1[g][k1], t);
                o[r] = t;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            dst[row * pstep] = o[r];
            vmax = fmaxf(vmax, fabsf(o[r]));
        }
    }
    if (a.out_absmax) publish_absmax(a.out_absmax, vmax, blockIdx.x * 4 + wave);
}

template <int KD>
static int launch_torgb_skip_kind(const HfagpTorgbSkipArgs* a, hipStream_t s) {
    const unsigned grid = (unsigned)(((long long)a->B * a->H * a->W) / 128);
    switch (a->Cout / 32) {
        case 1: torgb_skip_kernel<KD, 1><<<grid, 256, 0, s>>>(*a); break;
        case 2: torgb_skip_kernel<KD, 2><<<grid, 256, 0, s>>>(*a); break;
        case 3: torgb_skip_kernel<KD, 3><<<grid, 256, 0, s>>>(*a); break;
        default: torgb_skip_kernel<KD, 4><<<grid, 256, 0, s>>>(*a); break;
    }
    return check_launch("torgb_skip_fwd");
}

}  // namespace hfagp

using namespace hfagp;

extern "C" int hfagp_torgb_skip_fwd(const HfagpTorgbSkipArgs* a, void* stream) {
    HFAGP_REQUIRE(a && a->x && a->wt && a->styles && a->bias && a->img_out, HFAGP_EBADARG, "torgb_skip: null pointer");
    HFAGP_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0, HFAGP_EBADARG, "torgb_skip: bad dims");
    HFAGP_REQUIRE(a->Cout % 32 == 0 && a->Cout <= 128 && a->Cin % CKB == 0 && a->Cin <= 512 && a->W % 32 == 0 &&
                      (a->H * a->W) % 128 == 0 && (!a->img_in || (a->H % 2 == 0)),
                  HFAGP_EUNSUPPORTED, "torgb_skip: Cout=%d (multiple of 32, <= 128), Cin=%d (multiple of 16, <= 512), W=%d "
                                      "(multiple of 32), H*W=%d (multiple of 128); use hfagp_modconv_fwd + hfagp_skip_upsample_add",
                  a->Cout, a->Cin, a->W, a->H * a->W);
    HFAGP_REQUIRE(!a->plane_major || a->Cout == 96, HFAGP_EUNSUPPORTED, "torgb_skip: plane_major needs Cout = 96 (3 planes x 32)");
    hipStream_t s = (hipStream_t)stream;
    switch (kind_of(a->precision)) {
        case 1: return launch_torgb_skip_kind<1>(a, s);
        case 2: return launch_torgb_skip_kind<2>(a, s);
        case 3: return launch_torgb_skip_kind<3>(a, s);
        case 4: return launch_torgb_skip_kind<4>(a, s);
        case 5: return launch_torgb_skip_kind<4>(a, s);     // F16X2: the toRGB products stay fp32-class (same weight image)
        default: break;
    }
    set_error("torgb_skip: precision %d has no 16-bit weight image", a->precision);
    return HFAGP_EBADARG;
}

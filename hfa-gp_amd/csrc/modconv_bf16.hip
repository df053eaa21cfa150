// Modulated convolution on v_mfma_f32_32x32x16_bf16 with SPLIT operands (gfx950): every fp32 operand is the sum
// of NP bf16 parts (NP = 2: hi + lo, NP = 3: hi + mid + lo) and a product is the sum of the part products whose
// weight is above 2^-16 (NP = 2: hi.hi + lo.hi + hi.lo) or 2^-24 (NP = 3: six products), accumulated in fp32.
// The bf16 matrix pipe is 16x the fp32 one, so BF16X3 has 5.3x and BF16X6 2.7x the MFMA ceiling of the exact
// kernel (modconv.hip) at a relative product error of ~2^-16 / ~2^-23.
//
//   same implicit GEMM as modconv.hip: M = 8x16 output positions, N = 128 output channels, K = taps x Cin in
//   chunks of 16 channels (= one MFMA K step).  The activations are read as fp32, scaled by the style, split and
//   written to LDS as bf16 part images [part][position][16 ch] (32 B per position, the two 16-B halves XOR-swizzled
//   by bit 3 of the position so that a ds_read_b128 of 16 consecutive positions touches 16 distinct 16-B slots).
//   The weights are pre-split (hfagp_weight_prep_split) into [part][tap][Cin/8][Cout][8], which is the B-operand
//   fragment order: a lane's 8 K values are one 16-B load / one ds_read_b128.
//   A K chunk is consumed in steps of up to 3 taps; the B image of the next step and the A patch of the next chunk
//   are prefetched into registers under the MFMAs and written to the other LDS buffer before the step's barrier.
#include <type_traits>
#include "modconv_plan.h"

namespace hfagp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int CKB = 16;          // channels per K chunk
constexpr int TS = 3;            // taps per pipeline step
constexpr int LPWB = PW + 2;     // LDS row pitch of the patch in positions
constexpr int BNB = 128;         // output channels per block

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// v (4 floats) -> NP x 4 bf16 (two dwords per part)
template <int NP>
__device__ __forceinline__ void split4(float4 v, uint2 (&out)[NP]) {
    float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const unsigned lo = pack_bf16(r[0], r[1]), hi = pack_bf16(r[2], r[3]);
        out[p] = make_uint2(lo, hi);
        if (p + 1 < NP) {
            r[0] -= __builtin_bit_cast(float, lo << 16);
            r[1] -= __builtin_bit_cast(float, lo & 0xffff0000u);
            r[2] -= __builtin_bit_cast(float, hi << 16);
            r[3] -= __builtin_bit_cast(float, hi & 0xffff0000u);
        }
    }
}

template <int NP, int TM>
__global__ void __launch_bounds__(256, 2) modconv_bf16_kernel(const ConvParams p) {
    constexpr int TN = 2, WN = 2, BM = 2 * TM * 32, PH = BM / PW;
    constexpr int APOS = (PH + 2) * LPWB;                 // positions of the staged patch
    constexpr int A_PART = APOS * 32, A_BUF = NP * A_PART;
    constexpr int B_TAP = NP * 2 * BNB * 16, B_BUF = TS * B_TAP;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    char* As = lds_raw;                                   // [2][NP][APOS][32 B]
    char* Bs = lds_raw + 2 * A_BUF;                       // [2][TS][NP][2][128][16 B]

    const Phase& ph = p.phase[blockIdx.y];
    unsigned id = blockIdx.x;
    const int tn_blk = id % p.tiles_n; id /= p.tiles_n;
    const int tw = id % p.tiles_w;     id /= p.tiles_w;
    const int th = id % p.tiles_h;     id /= p.tiles_h;
    const int b = id % p.B;            id /= p.B;
    const int ks = id;
    const int m0 = th * PH, n0 = tw * PW, co0 = tn_blk * BNB;
    if (m0 >= ph.mh || n0 >= ph.mw) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5, l31 = lane & 31;

    const int c_begin = (int)(((long long)p.nchunks * ks) / p.ksplit);
    const int c_end = (int)(((long long)p.nchunks * (ks + 1)) / p.ksplit);

    // ---- A staging: float4 (4 channels of one position) per slot, 4 slots per position.  Loads are branch-free:
    // a slot outside the image (zero padding) or past the patch reads element 0 and is multiplied by 0.
    const int npatch = p.ph * p.pw;
    constexpr int A_PER_T = ((PH + 2) * (PW + 2) * 4 + 255) / 256;
    float4 ra[A_PER_T], rs[A_PER_T];
    const float* xb = p.x + ph.in_off + (long long)b * p.x_batch_stride;
    const float* sb = p.styles ? p.styles + (size_t)b * p.Cin : nullptr;
    int aoff[A_PER_T], lds_a[A_PER_T];
    float amask[A_PER_T];
#pragma unroll
    for (int k = 0; k < A_PER_T; ++k) {
        const int idx = tid + k * 256;
        aoff[k] = 0;
        amask[k] = 0.f;
        lds_a[k] = -1;
        if (idx < npatch * 4) {
            const int pix = idx >> 2, q = idx & 3;
            const int pos = (pix / p.pw) * LPWB + pix % p.pw;
            lds_a[k] = pos * 32 + ((((q >> 1) ^ (pos >> 3)) & 1) << 4) + ((q & 1) << 3);
            const int iy = m0 + p.dymin + pix / p.pw, ix = n0 + p.dxmin + pix % p.pw;
            if (iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) {
                aoff[k] = (iy * p.in_w + ix) * p.Cin + 4 * q;      // < 2^31: one image of the batch
                amask[k] = 1.f;
            }
        }
    }
    auto load_a = [&](int chunk) __attribute__((always_inline)) {
        const int c0 = chunk * CKB;
#pragma unroll
        for (int k = 0; k < A_PER_T; ++k) {
            ra[k] = *reinterpret_cast<const float4*>(xb + aoff[k] + c0);
            float4 sv = make_float4(1.f, 1.f, 1.f, 1.f);
            if (sb) sv = *reinterpret_cast<const float4*>(sb + c0 + 4 * ((tid + k * 256) & 3));
            rs[k] = sv;
        }
    };
    auto store_a = [&](int buf) __attribute__((always_inline)) {
        char* dst = As + buf * A_BUF;
#pragma unroll
        for (int k = 0; k < A_PER_T; ++k) {
            if (lds_a[k] < 0) continue;
            const float m = amask[k];
            uint2 parts[NP];
            split4<NP>(make_float4(ra[k].x * (rs[k].x * m), ra[k].y * (rs[k].y * m), ra[k].z * (rs[k].z * m),
                                   ra[k].w * (rs[k].w * m)), parts);
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(dst + q * A_PART + lds_a[k]) = parts[q];
        }
    };

    // ---- B staging: one step = up to TS taps x NP parts x [2 k-groups][128 co] 16-B slots; thread = (kg, co)
    u32x4 rb[TS * NP];
    const u32x4* wb = reinterpret_cast<const u32x4*>(p.wt);
    const int cq8 = p.Cin >> 3;
    const int part_stride = p.wtaps * cq8 * p.Cout;                           // uint4 per part
    const int bthread = (tid >> 7) * p.Cout + co0 + (tid & 127);
    int wtap[MAXTAPS];                                                        // tap table -> registers, once
#pragma unroll
    for (int t = 0; t < MAXTAPS; ++t) wtap[t] = t < ph.ntaps ? ph.widx[t] * cq8 * p.Cout : 0;

    // ---- per-lane fragment addresses (bytes inside one part image), one per (tap, M tile)
    int aaddr[MAXTAPS][TM];
#pragma unroll
    for (int t = 0; t < MAXTAPS; ++t) {
        const int tpos = t < ph.ntaps ? (ph.dy[t] - p.dymin) * LPWB + (ph.dx[t] - p.dxmin) : 0;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int pidx = (wm * TM + tm) * 32 + l31;
            const int pos = (pidx >> 4) * LPWB + (pidx & 15) + tpos;
            aaddr[t][tm] = pos * 32 + (((h ^ (pos >> 3)) & 1) << 4);
        }
    }
    int bcol[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bcol[tn] = (h * BNB + (wn * TN + tn) * 32 + l31) * 16;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    // part products in the order they are issued: (A part, B part)
    constexpr int NPROD = NP == 2 ? 3 : 6;
    constexpr int PA[6] = {0, 1, 0, 1, 2, 0};
    constexpr int PB[6] = {0, 0, 1, 1, 0, 2};

    // the K loop for a compile-time tap count NT (9: 3x3, 4/2/1: the phases of the stride-2 transposed conv and
    // the 1x1 conv): straight-line steps of up to TS taps, so the LDS reads of a tap are scheduled under the MFMAs
    // of the tap before
    auto run = [&](auto nt_tag) __attribute__((always_inline)) {
        constexpr int NT = decltype(nt_tag)::value;
        constexpr int NSTEPS = (NT + TS - 1) / TS;
        auto load_b = [&](int chunk, auto step_tag) __attribute__((always_inline)) {
            constexpr int S = decltype(step_tag)::value;
            constexpr int NJ = NT - S * TS < TS ? NT - S * TS : TS;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    rb[j * NP + q] = wb[q * part_stride + wtap[S * TS + j] + chunk * 2 * p.Cout + bthread];
        };
        auto store_b = [&](int buf, auto step_tag) __attribute__((always_inline)) {
            constexpr int S = decltype(step_tag)::value;
            char* dst = Bs + buf * B_BUF + tid * 16;
            constexpr int NJ = NT - S * TS < TS ? NT - S * TS : TS;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < NP; ++q) *reinterpret_cast<u32x4*>(dst + (j * NP + q) * 4096) = rb[j * NP + q];
        };
        auto step = [&](int c, int abuf, int& bbuf, auto step_tag) __attribute__((always_inline)) {
            constexpr int S = decltype(step_tag)::value;
            store_b(bbuf, step_tag);
            __syncthreads();
#ifndef HFAGP_DIAG_NOLOAD
            if constexpr (S + 1 < NSTEPS) load_b(c, std::integral_constant<int, S + 1>{});
            else if (c + 1 < c_end) load_b(c + 1, std::integral_constant<int, 0>{});
            if (S == 0 && c + 1 < c_end) load_a(c + 1);
#endif
            const char* Ac = As + abuf * A_BUF;
            const char* Bc = Bs + bbuf * B_BUF;
            constexpr int NJ = NT - S * TS < TS ? NT - S * TS : TS;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int t = S * TS + j;
                bf16x8 af[TM][NP], bfr[TN][NP];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int q = 0; q < NP; ++q)
                        af[tm][q] = *reinterpret_cast<const bf16x8*>(Ac + q * A_PART + aaddr[t][tm]);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int q = 0; q < NP; ++q)
                        bfr[tn][q] = *reinterpret_cast<const bf16x8*>(Bc + (j * NP + q) * 4096 + bcol[tn]);
#ifdef HFAGP_DIAG_NOMFMA
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                        for (int q = 0; q < NP; ++q) {
                            acc[tm][tn][q] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, af[tm][q])[0]);
                            acc[tm][tn][q + 4] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, bfr[tn][q])[1]);
                        }
                continue;
#endif
#pragma unroll
                for (int pr = 0; pr < NPROD; ++pr)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[tm][PA[pr]], bfr[tn][PB[pr]],
                                                                                  acc[tm][tn], 0, 0, 0);
            }
            bbuf ^= 1;
        };
        int bbuf = 0;
        if (c_begin < c_end) { load_a(c_begin); load_b(c_begin, std::integral_constant<int, 0>{}); }
        for (int c = c_begin; c < c_end; ++c) {
            const int abuf = (c - c_begin) & 1;
            store_a(abuf);
            step(c, abuf, bbuf, std::integral_constant<int, 0>{});
            if constexpr (NSTEPS > 1) step(c, abuf, bbuf, std::integral_constant<int, 1>{});
            if constexpr (NSTEPS > 2) step(c, abuf, bbuf, std::integral_constant<int, 2>{});
        }
    };
    switch (ph.ntaps) {
        case 9: run(std::integral_constant<int, 9>{}); break;
        case 4: run(std::integral_constant<int, 4>{}); break;
        case 2: run(std::integral_constant<int, 2>{}); break;
        default: run(std::integral_constant<int, 1>{}); break;
    }

    // ---- epilogue.  C/D layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* out = p.out + (size_t)(ks * p.nslab + ph.slab) * p.slab;
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
            float nz[16];
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
                out[(((size_t)b * p.Ho + oy) * p.Wo + ox) * p.Cout + co] = v;
            }
        }
    }
}

template <int NP, int TM>
static size_t bf16_lds_bytes() {
    constexpr int PH = 2 * TM * 32 / PW;
    return (size_t)2 * NP * (PH + 2) * LPWB * 32 + (size_t)2 * TS * NP * 2 * BNB * 16;
}

int launch_modconv_bf16(const HfagpModconvArgs* a, Plan& pl, hipStream_t s) {
    HFAGP_REQUIRE(a->Cin % CKB == 0 && a->Cout % BNB == 0, HFAGP_EUNSUPPORTED,
                  "modconv (split bf16): Cin=%d must be a multiple of %d and Cout=%d of %d", a->Cin, CKB, a->Cout, BNB);
    HFAGP_REQUIRE(pl.bn == BNB && pl.bm == 128, HFAGP_EUNSUPPORTED, "modconv (split bf16): unexpected plan");
    static bool attr = false;           // both images exceed the 64 KB default dynamic-LDS limit
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(modconv_bf16_kernel<2, 2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)bf16_lds_bytes<2, 2>());
        hipFuncSetAttribute(reinterpret_cast<const void*>(modconv_bf16_kernel<3, 2>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)bf16_lds_bytes<3, 2>());
        attr = true;
    }
    if (a->precision == HFAGP_PREC_BF16X3) {
        modconv_bf16_kernel<2, 2><<<pl.grid, 256, bf16_lds_bytes<2, 2>(), s>>>(pl.p);
    } else if (a->precision == HFAGP_PREC_BF16X6) {
        modconv_bf16_kernel<3, 2><<<pl.grid, 256, bf16_lds_bytes<3, 2>(), s>>>(pl.p);
    } else {
        set_error("modconv: unknown precision %d", a->precision);
        return HFAGP_EBADARG;
    }
    return check_launch("modconv_fwd (split bf16)");
}

// weight [Cout][Cin][taps] -> wb [nparts][taps][Cin/8][Cout][8] bf16; thread = (tap, ci group, co)
__global__ void __launch_bounds__(256) weight_prep_split_kernel(const float* __restrict__ w, uint4* __restrict__ wb,
                                                                int Cout, int Cin, int taps, int nparts) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cq8 = Cin >> 3;
    const long long n = (long long)taps * cq8 * Cout;
    if (idx >= n) return;
    const int co = (int)(idx % Cout);
    const int g = (int)((idx / Cout) % cq8);
    const int t = (int)(idx / ((long long)Cout * cq8));
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = w[((size_t)co * Cin + 8 * g + e) * taps + t];
    for (int q = 0; q < nparts; ++q) {
        unsigned u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            u[e] = pack_bf16(r[2 * e], r[2 * e + 1]);
            r[2 * e] -= __builtin_bit_cast(float, u[e] << 16);
            r[2 * e + 1] -= __builtin_bit_cast(float, u[e] & 0xffff0000u);
        }
        wb[q * n + idx] = make_uint4(u[0], u[1], u[2], u[3]);
    }
}

}  // namespace hfagp

using namespace hfagp;

extern "C" int hfagp_weight_prep_split(const float* weight, void* wb, int32_t Cout, int32_t Cin, int32_t taps,
                                       int32_t nparts, void* stream) {
    HFAGP_REQUIRE(weight && wb, HFAGP_EBADARG, "weight_prep_split: null pointer");
    HFAGP_REQUIRE(Cin % 8 == 0 && Cout > 0 && (taps == 1 || taps == 9) && (nparts == 2 || nparts == 3),
                  HFAGP_EUNSUPPORTED, "weight_prep_split: Cin=%d must be a multiple of 8, taps=%d in {1,9}, nparts=%d in {2,3}",
                  Cin, taps, nparts);
    const long long n = (long long)taps * (Cin / 8) * Cout;
    weight_prep_split_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        weight, reinterpret_cast<uint4*>(wb), Cout, Cin, taps, nparts);
    return check_launch("weight_prep_split");
}

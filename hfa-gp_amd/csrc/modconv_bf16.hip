// Modulated convolution on the 16-bit matrix pipe (v_mfma_f32_32x32x16_{f16,bf16}, gfx950) with SPLIT operands:
// every fp32 operand is the sum of 16-bit parts, each the round-to-nearest value of the residual left by the parts
// before it, and a product is the sum of the part products above the target weight, accumulated in fp32.  Operand
// kinds (template parameter KD):
//   KD = 4  F16X3 (default)  fp16 hi + lo (11 + 11 mantissa bits), hi.hi + lo.hi + hi.lo:      ~2^-22 per product
//   KD = 2  BF16X3           bf16 hi + lo ( 8 +  8 bits),          the same three products:      ~2^-16
//   KD = 3  BF16X6           bf16 hi + mid + lo,                   six products:                 ~2^-23
//   KD = 1  F16              one fp16 rounding,                    ONE product (EG3D's fp16 blocks): ~2^-11
// The 16-bit pipe is 16x the fp32 one, so the 3-product kinds have 5.3x the MFMA ceiling of the exact kernel
// (modconv.hip).  The fp16 kinds carry a range guard (style_range_guard) and a saturating split.
//
//   same implicit GEMM as modconv.hip: M = 8x16 output positions, N = 128 output channels, K = taps x Cin in
//   chunks of 16 channels (= one MFMA K step).  The activations are read as fp32, scaled by the style, split and
//   written to LDS as part images [part][position][16 ch + 16 B pad] (48-B pitch, row pitch 32 positions: every
//   16-lane group of a ds_read_b128 covers 16 distinct 16-B slots).
//   The weights are pre-split (hfagp_weight_prep_prec) into [part][tap][Cin/8][Cout][8], which is the B-operand
//   fragment order: a lane's 8 K values are one 16-B load, straight from L2 into the fragment registers through a
//   ring that stays 3 (F16: 6) taps ahead of the MFMAs.  The patch of the next chunk is converted under the last
//   taps of the current one into the other LDS buffer: one barrier per chunk.
#include <type_traits>
#include "conv16_common.h"

namespace hfagp {

// IO: fp16 STORAGE of the activations (hfagp.h x_f16 / y_f16; KD = 1 only): bit 0 = x is fp16 (staging copies the halves
// and applies the style with packed fp16 multiplies), bit 1 = y is written as fp16
template <int KD, int TM, int NTAPS, int IO = 0>
__global__ void __launch_bounds__(256, 2) modconv_bf16_kernel(const ConvParams p, const int phase0) {
    constexpr int NP = kind_parts_a(KD), NPB = kind_parts(KD);    // parts of the activations (LDS patch) / of the weight image
    constexpr bool F16 = kind_f16(KD);
    constexpr bool XH = (IO & 1) != 0, YH = (IO & 2) != 0;
    constexpr int XB = XH ? 2 : 4;                       // bytes per input element
    static_assert(IO == 0 || KD == 1, "fp16 storage goes with the single-pass fp16 arithmetic");
#ifndef HFAGP_WAVES_N
#define HFAGP_WAVES_N 2
#endif
#ifndef HFAGP_LOADA_EARLY
#define HFAGP_LOADA_EARLY 1
#endif
#ifndef HFAGP_B_EARLY
#define HFAGP_B_EARLY 1
#endif
    // wave grid WM x WN over the 128 x 128 block tile; TM here is the M tiles per wave for WN = 2
    constexpr int WN = HFAGP_WAVES_N, WM = 4 / WN, TN = 4 / WN, TMW = 4 / WM, BM = 128, PH = BM / PW;
    static_assert(TM == 2, "block tile is 128 positions");
    constexpr int LPWB = RowPitch<NP>::value;
    constexpr int APOS = (PH + 2) * LPWB;                 // positions of the staged patch
    constexpr int A_PART = APOS * APITCH, A_BUF = NP * A_PART;
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    char* As = lds_raw;                                   // [2][NP][APOS][48 B]
    float* Ss = reinterpret_cast<float*>(lds_raw + 2 * A_BUF);   // [Cin] styles of this sample (or ones)

    // NTAPS = 0: the adjoint of the up-sampling conv (mode CONVS2_BWD) MERGED — the four parity phases (4 | 2 | 2 | 1 taps, each
    // reading its own parity image of the y_t gradient) run one after the other in this block into ONE accumulator set, so the
    // result is written once (round 3 launched the three tap counts separately, each into its own slab, and summed the slabs
    // in splitk_epilogue_kernel: 4 launches and 4 x the output traffic per layer).  All four phases share the output grid.
    constexpr bool MERGED_S2 = NTAPS == 0;
    const Phase& ph = p.phase[MERGED_S2 ? 0 : phase0 + blockIdx.y];   // every phase of one launch has NTAPS taps
    // (readfirstlane: the divisions by run-time values are done on the vector ALU; without it every index derived
    // from the block coordinates stays in VGPRs and the uniform address arithmetic of the K loop — chunk x Cout
    // products, clamps — is issued as quarter-rate vector multiplies between the MFMAs)
    unsigned id = p.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int tn_blk = __builtin_amdgcn_readfirstlane(id % p.tiles_n); id /= p.tiles_n;
    const int tw = __builtin_amdgcn_readfirstlane(id % p.tiles_w);     id /= p.tiles_w;
    const int th = __builtin_amdgcn_readfirstlane(id % p.tiles_h);     id /= p.tiles_h;
    const int b = __builtin_amdgcn_readfirstlane(id % p.B);            id /= p.B;
    const int ks = __builtin_amdgcn_readfirstlane(id);
    const int m0 = th * PH, n0 = tw * PW, co0 = tn_blk * BNB;
    if (m0 >= ph.mh || n0 >= ph.mw) return;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5, l31 = lane & 31;

    const int c_begin = __builtin_amdgcn_readfirstlane((int)(((long long)p.nchunks * ks) / p.ksplit));
    const int c_end = __builtin_amdgcn_readfirstlane((int)(((long long)p.nchunks * (ks + 1)) / p.ksplit));

    // ---- A staging: float4 (4 channels of one position) per slot, 4 slots per position.  Everything is
    // branch-free: a slot outside the image (zero padding) reads element 0 and is multiplied by 0, a thread past
    // the end of the patch repeats the last slot (same value to the same LDS address), and addresses are a
    // uniform base + a 32-bit per-lane byte offset (global_load with an SGPR base: no 64-bit VALU address math).
    const int npatch = p.ph * p.pw;
    constexpr int A_PER_T = ((PH + 2) * (PW + 2) * 4 + 255) / 256;
    static_assert(A_PER_T == 3, "the staging schedule below is written for three slots per thread");
    float4 ra[A_PER_T];
    const char* xb = reinterpret_cast<const char*>(p.x) + (ph.in_off + (long long)b * p.x_batch_stride) * XB;   // (MERGED_S2: per phase)
    for (int i = tid; i < p.Cin; i += 256) Ss[i] = p.styles ? p.styles[(size_t)b * p.Cin + i] : 1.f;
    float sback = 1.f, sdown = 1.f;                      // 2^e, 2^-e of the fp16 range guard (1 for the bf16 kinds)
    if constexpr (F16) sdown = style_range_guard(p.styles ? p.styles + (size_t)b * p.Cin : nullptr, p.Cin, lane, &sback, p.x_absmax);
    unsigned aoff[A_PER_T];
    int lds_a[A_PER_T], soff[A_PER_T];
    float amask[A_PER_T];
#pragma unroll
    for (int k = 0; k < A_PER_T; ++k) {
        const int idx = min(tid + k * 256, npatch * 4 - 1);
        const int pix = idx >> 2, q = idx & 3;
        lds_a[k] = ((pix / p.pw) * LPWB + pix % p.pw) * APITCH + 8 * q;
        const int iy = m0 + p.dymin + pix / p.pw, ix = n0 + p.dxmin + pix % p.pw;
        const bool inside = iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
        aoff[k] = inside ? (unsigned)((iy * p.in_w + ix) * p.Cin + 4 * q) * (unsigned)XB : 0u;   // bytes, < 2^32 per image
        amask[k] = inside ? sdown : 0.f;          // zero padding and the fp16 range guard in one factor
        soff[k] = 4 * q;
    }
    auto load_a = [&](int chunk) __attribute__((always_inline)) {
        const char* xc = xb + (long long)chunk * (CKB * XB);
#pragma unroll
        for (int k = 0; k < A_PER_T; ++k) {
            if constexpr (XH) {
                const uint2 u = *reinterpret_cast<const uint2*>(xc + aoff[k]);
                ra[k].x = __builtin_bit_cast(float, u.x);
                ra[k].y = __builtin_bit_cast(float, u.y);
            } else {
                ra[k] = *reinterpret_cast<const float4*>(xc + aoff[k]);
            }
        }
    };
    // slot K of the staged patch of `chunk`: scale by the style, split, write the parts to LDS buffer BUF
    auto store_a = [&](int chunk, auto buf_tag, auto k_tag) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_tag)::value, k = decltype(k_tag)::value;
        {
            const float m = amask[k];
            const float4 sv = *reinterpret_cast<const float4*>(Ss + chunk * CKB + soff[k]);
            uint2 parts[NP];
            if constexpr (XH) {
                // fp16 storage: the halves are the operand already; the style (|s| <= 1 after the range guard, |x| <= the
                // layer's clamp) goes on with two packed fp16 multiplies, as EG3D's fp16 blocks do
                const f32x2 s01 = {sv.x * m, sv.y * m}, s23 = {sv.z * m, sv.w * m};
                const f16x2 x01 = __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, ra[k].x));
                const f16x2 x23 = __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, ra[k].y));
                parts[0] = make_uint2(__builtin_bit_cast(unsigned, x01 * __builtin_convertvector(s01, f16x2)),
                                      __builtin_bit_cast(unsigned, x23 * __builtin_convertvector(s23, f16x2)));
            } else
            split4<KD>(make_float4(ra[k].x * (sv.x * m), ra[k].y * (sv.y * m), ra[k].z * (sv.z * m),
                                   ra[k].w * (sv.w * m)), parts);     // (F16X2: one saturating fp16 part)
#pragma unroll
            for (int q = 0; q < NP; ++q)
                *reinterpret_cast<uint2*>(As + BUF * A_BUF + q * A_PART + lds_a[k]) = parts[q];
        }
    };

    // ---- B operand: straight from global/L2 into the fragment registers (no LDS, no barrier): a lane's fragment
    // is 16 contiguous bytes of the split image and lanes 0-31 / 32-63 of a fragment read two contiguous 512-B
    // runs, so the loads are full 128-B lines.  A ring of RB (2..4) fragment sets keeps the loads RB items ahead of the
    // MFMAs that consume them (item = one tap of one K chunk).
    const char* wb = reinterpret_cast<const char*>(p.wt);
    const int cq8 = p.Cin >> 3;
    // Cout = 96 (toRGB) runs on the same 128-wide tile: the lanes of its last 32 columns read the first 32 entries of
    // the NEXT image row (the caller pads the buffer by 512 B for the very last row), their accumulators are never
    // stored.  (A separate padded pitch for the image cost 66 VGPRs in the 9-tap kernel: spills.)
    const int part_stride = p.wtaps * cq8 * p.Cout;                           // uint4 per part
    unsigned bth[TN];                                                         // per-lane byte offset of a fragment
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bth[tn] = (unsigned)(h * p.Cout + co0 + (wn * TN + tn) * 32 + l31) * 16u;
    int wtap[MAXTAPS];                                                        // tap table -> registers, once
#pragma unroll
    for (int t = 0; t < MAXTAPS; ++t) wtap[t] = t < ph.ntaps ? ph.widx[t] * cq8 * p.Cout : 0;

    // ---- per-lane fragment address of each M tile (bytes inside one part image, tap (0,0)); a tap adds the
    // uniform offset toff[t] (for the 3x3 modes the taps are sorted by (dy, dx), see make_plan, so that the offset
    // is a compile-time immediate of the ds_read)
    int apos[TMW];
#pragma unroll
    for (int tm = 0; tm < TMW; ++tm) {
        const int pidx = (wm * TMW + tm) * 32 + l31;
        apos[tm] = ((pidx >> 4) * LPWB + (pidx & 15)) * APITCH + 16 * h;
    }
    int toff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
        toff[t] = t < ph.ntaps ? ((ph.dy[t] - p.dymin) * LPWB + (ph.dx[t] - p.dxmin)) * APITCH : 0;
    // (MERGED_S2) switch the phase-dependent state — input image, weight taps, LDS tap offsets — to parity phase q
    auto enter_phase = [&](const Phase& q) __attribute__((always_inline)) {
        xb = reinterpret_cast<const char*>(p.x) + (q.in_off + (long long)b * p.x_batch_stride) * XB;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            wtap[t] = t < q.ntaps ? q.widx[t] * cq8 * p.Cout : 0;
            toff[t] = t < q.ntaps ? ((q.dy[t] - p.dymin) * LPWB + (q.dx[t] - p.dxmin)) * APITCH : 0;
        }
    };

    f32x16 acc[TMW][TN];
#pragma unroll
    for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    // part products in the order they are issued: (A part, B part)
    constexpr int NPROD = kind_nprod(KD);
    constexpr int PA[6] = {kind_pa(KD, 0), kind_pa(KD, 1), kind_pa(KD, 2), kind_pa(KD, 3), kind_pa(KD, 4), kind_pa(KD, 5)};
    constexpr int PB[6] = {kind_pb(KD, 0), kind_pb(KD, 1), kind_pb(KD, 2), kind_pb(KD, 3), kind_pb(KD, 4), kind_pb(KD, 5)};

    // the K loop for a compile-time tap count NT (9: 3x3, 4/2/1: the phases of the stride-2 transposed conv and
    // the 1x1 conv).  U chunks are unrolled so that U*NT is a multiple of the ring size: every item then has a
    // compile-time ring slot and the whole group is straight-line code.
    auto run = [&](auto nt_tag) __attribute__((always_inline)) {
        constexpr int NT = decltype(nt_tag)::value;
        constexpr bool EARLY_A = HFAGP_LOADA_EARLY && NT == 9;
#ifndef HFAGP_RB9
#define HFAGP_RB9 3
#endif
        // ring size: divides U*NT (the single-pass fp16 mode has a third of the MFMA time per item: deeper ring)
        constexpr int RB = NT == 9 ? (NPROD == 1 ? 6 : HFAGP_RB9) : NT == 4 ? 4 : 2;
        constexpr int U = 2;                                   // chunk pairs: chunk parity = A buffer = compile time
        u32x4 bq[RB][TN][NPB];
        // loads of item (chunk c, tap t) into ring slot `slot`; c is clamped so that the look-ahead past the last
        // chunk re-reads valid memory instead of branching
        auto issue_b = [&](int c, auto t_tag, auto slot_tag) __attribute__((always_inline)) {
            constexpr int T = decltype(t_tag)::value, SL = decltype(slot_tag)::value;
            const int cc = min(c, c_end - 1);
#pragma unroll
            for (int q = 0; q < NPB; ++q) {
                const char* base = wb + (long long)(q * part_stride + wtap[T] + cc * 2 * p.Cout) * 16;   // uniform
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) bq[SL][tn][q] = *reinterpret_cast<const u32x4*>(base + bth[tn]);
            }
        };
        u32x4 af[2][TMW][NP];                                // A fragments of the current and the next tap
        auto read_a = [&](auto u_tag, auto t_tag) __attribute__((always_inline)) {
            constexpr int UU = decltype(u_tag)::value, T = decltype(t_tag)::value;
            const char* Ac = As + UU * A_BUF;               // chunk parity = LDS buffer: immediate offsets
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    if constexpr (NT == 9)
                        af[T & 1][tm][q] = *reinterpret_cast<const u32x4*>(
                            Ac + q * A_PART + ((T / 3) * LPWB + T % 3) * APITCH + apos[tm]);
                    else
                        af[T & 1][tm][q] = *reinterpret_cast<const u32x4*>(Ac + q * A_PART + toff[T & 3] + apos[tm]);
                }
        };
        // The patch of chunk c+1 is converted and written to the OTHER LDS buffer inside the last A_PER_T taps of
        // chunk c, one slot per tap and in the same scheduling region as that tap's MFMAs, so that the VALU work
        // of the conversion is issued between MFMAs instead of in front of the barrier.
        auto item = [&](int c, auto u_tag, auto t_tag) __attribute__((always_inline)) {
            constexpr int UU = decltype(u_tag)::value, T = decltype(t_tag)::value;
            constexpr int SL = (UU * NT + T) % RB;
            // LDS reads of the next tap go out before this tap's MFMAs (the last tap of a chunk has no successor in
            // this buffer: the next chunk's patch is published by the barrier in between)
            // (sched_barrier: without it the scheduler sinks every load to just before its first use to save
            // registers, i.e. it undoes the look-ahead)
            if constexpr (T + 1 < NT) read_a(u_tag, std::integral_constant<int, T + 1>{});
#if HFAGP_B_EARLY
            {   // B fragments of the item RB-1 ahead, into the slot the PREVIOUS item has just finished with: issued in
                // front of this item's MFMAs, so the youngest load at the loop's back edge (where hipcc drains vmcnt
                // to 0) is a whole item old instead of brand new
                constexpr int TE = (T + RB - 1) % NT, DCE = (T + RB - 1) / NT, SLE = (UU * NT + T + RB - 1) % RB;
                issue_b(c + DCE, std::integral_constant<int, TE>{}, std::integral_constant<int, SLE>{});
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pr = 0; pr < NPROD; ++pr)
#pragma unroll
                for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = mfma16<F16>(af[T & 1][tm][PA[pr]], bq[SL][tn][PB[pr]], acc[tm][tn]);
#pragma unroll
            for (int k = 0; k < A_PER_T; ++k) {
                const int tk = NT - A_PER_T + k < 0 ? 0 : NT - A_PER_T + k;      // tap that carries slot k
                if (tk == T) {
                    if (k == 0) store_a(min(c + 1, c_end - 1), std::integral_constant<int, 1 - UU>{}, std::integral_constant<int, 0>{});
                    if (k == 1) store_a(min(c + 1, c_end - 1), std::integral_constant<int, 1 - UU>{}, std::integral_constant<int, 1>{});
                    if (k == 2) store_a(min(c + 1, c_end - 1), std::integral_constant<int, 1 - UU>{}, std::integral_constant<int, 2>{});
                }
            }
            // refill the slot with the item RB ahead
            constexpr int T2 = (T + RB) % NT, DC = (T + RB) / NT;
#if !HFAGP_B_EARLY
            issue_b(c + DC, std::integral_constant<int, T2>{}, std::integral_constant<int, SL>{});
#endif
            // 9 taps: the patch of chunk c+1 is fetched at the FIRST tap of chunk c and converted under its last taps, so
            // nothing but B fragments is in flight at the loop's back edge, where hipcc drains vmcnt to 0 (its wait-count
            // analysis is conservative at loop headers) — with the fetch at the last tap that drain waited for HBM.
            // Fewer taps: fetch at the last tap for chunk c+2 (a whole chunk of cover).
            if constexpr (EARLY_A) {
                if constexpr (T == 0) load_a(min(c + 1, c_end - 1));
            } else {
                if constexpr (T == NT - 1) load_a(min(c + 2, c_end - 1));
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto chunk = [&](int c, auto u_tag) __attribute__((always_inline)) {
            __syncthreads();                                // publishes the patch of chunk c
            read_a(u_tag, std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
            item(c, u_tag, std::integral_constant<int, 0>{});
            if constexpr (NT > 1) item(c, u_tag, std::integral_constant<int, 1>{});
            if constexpr (NT > 2) {
                item(c, u_tag, std::integral_constant<int, 2>{});
                item(c, u_tag, std::integral_constant<int, 3>{});
            }
            if constexpr (NT > 4) {
                item(c, u_tag, std::integral_constant<int, 4>{});
                item(c, u_tag, std::integral_constant<int, 5>{});
                item(c, u_tag, std::integral_constant<int, 6>{});
                item(c, u_tag, std::integral_constant<int, 7>{});
                item(c, u_tag, std::integral_constant<int, 8>{});
            }
        };
        if (c_begin >= c_end) return;
        __syncthreads();                                    // styles are in LDS
        load_a(c_begin);
        store_a(c_begin, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        store_a(c_begin, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        store_a(c_begin, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
        if constexpr (!EARLY_A) load_a(min(c_begin + 1, c_end - 1));
        // prologue: the first RB items
        issue_b(c_begin + 0 / NT, std::integral_constant<int, 0 % NT>{}, std::integral_constant<int, 0>{});
        // (HFAGP_B_EARLY: the item itself issues the fragments RB-1 ahead, so the prologue stops one item short)
        constexpr int NPRO = HFAGP_B_EARLY ? RB - 1 : RB;
        if constexpr (NPRO > 1)
            issue_b(c_begin + 1 / NT, std::integral_constant<int, 1 % NT>{}, std::integral_constant<int, 1>{});
        if constexpr (NPRO > 2)
            issue_b(c_begin + 2 / NT, std::integral_constant<int, 2 % NT>{}, std::integral_constant<int, 2>{});
        if constexpr (NPRO > 3)
            issue_b(c_begin + 3 / NT, std::integral_constant<int, 3 % NT>{}, std::integral_constant<int, 3>{});
        if constexpr (NPRO > 4)
            issue_b(c_begin + 4 / NT, std::integral_constant<int, 4 % NT>{}, std::integral_constant<int, 4>{});
        if constexpr (NPRO > 5)
            issue_b(c_begin + 5 / NT, std::integral_constant<int, 5 % NT>{}, std::integral_constant<int, 5>{});
        static_assert((U * NT) % RB == 0, "ring slots must repeat every iteration");
        for (int cg = c_begin; cg < c_end; cg += U) {
            chunk(cg, std::integral_constant<int, 0>{});
            if (cg + 1 >= c_end) break;
            chunk(cg + 1, std::integral_constant<int, 1>{});
        }
    };
    if constexpr (MERGED_S2) {
        run(std::integral_constant<int, 4>{});            // parity (0,0): phase[0] is the state set up above
        enter_phase(p.phase[1]);
        run(std::integral_constant<int, 2>{});
        enter_phase(p.phase[2]);
        run(std::integral_constant<int, 2>{});
        enter_phase(p.phase[3]);
        run(std::integral_constant<int, 1>{});
    } else {
        run(std::integral_constant<int, NTAPS>{});
    }

    // ---- epilogue.  C/D layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5): the 16 registers of
    // a tile are 2 patch rows (r>>3) x columns 8*((r>>2)&1) + 4h + (r&3).  One 64-bit row pointer per (tile, patch
    // row); everything else is a 32-bit offset (the per-element 64-bit index products were ~8k VALU cycles per wave).
    float* out = p.out + (size_t)(ks * p.nslab + ph.slab) * p.slab;       // (YH: no split-K, p.out = y as fp16)
    const int cstep = ph.sx * p.Cout;                                     // elements between neighbouring columns
    float vmax = 0.f;                                                     // max |y| of this lane's stores (y_absmax)
    // fused toRGB (hfagp.h rgb_w / rgb_part): rgbp[position][r] collects y[co] * rgb_w[r][co] over this lane's channels
    constexpr bool RGB = NTAPS == 9 || NTAPS == 1;
    const bool do_rgb = RGB && p.fused && p.rgb_part != nullptr;
    float rgbp[RGB ? TMW * 16 * 3 : 1];
    if constexpr (RGB) {
#pragma unroll
        for (int i = 0; i < TMW * 16 * 3; ++i) rgbp[i] = 0.f;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int co = co0 + (wn * TN + tn) * 32 + l31;
        if (co >= p.Cout) continue;
        float rw3[3] = {0.f, 0.f, 0.f};
        if constexpr (RGB)
            if (do_rgb) {
#pragma unroll
                for (int r = 0; r < 3; ++r) rw3[r] = p.rgb_w[((size_t)b * 3 + r) * p.Cout + co];
            }
        float d = sback, bs = 0.f;
        if (p.fused) {
            if (p.dcoef) d = p.dcoef[(size_t)b * p.Cout + co] * sback;
            if (p.bias) bs = p.bias[co];
        }
#pragma unroll
        for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
            for (int rw = 0; rw < 2; ++rw) {
                const int m = m0 + 2 * (wm * TMW + tm) + rw;
                if (m >= ph.mh) continue;
                const int oy = ph.sy * m + ph.oy0;
                const size_t rowoff = (((size_t)b * p.Ho + oy) * p.Wo + ph.ox0) * p.Cout + co;
                float* rowp = out + rowoff;
                _Float16* rowh = reinterpret_cast<_Float16*>(p.out) + rowoff;
                const float* nrow = (p.fused && p.noise) ? p.noise + (size_t)oy * p.Wo + ph.ox0 : nullptr;
                float nz[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int n = min(n0 + 8 * (q >> 2) + 4 * h + (q & 3), ph.mw - 1);
                    nz[q] = nrow ? nrow[ph.sx * n] * p.noise_strength : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int n = n0 + 8 * (q >> 2) + 4 * h + (q & 3);
                    if (n >= ph.mw) continue;
                    float v = acc[tm][tn][8 * rw + q];
                    if (p.fused) v = lrelu_gain_clamp(v * d + bs + nz[q], p.act, p.alpha, p.gain, p.clamp);
                    else if constexpr (F16) v *= sback;
                    vmax = fmaxf(vmax, fabsf(v));
                    if (p.out) {                 // (NULL: the caller only wants the fused toRGB sums — last SR layer, forward only)
                        if constexpr (YH) rowh[n * cstep] = (_Float16)v;
                        else rowp[n * cstep] = v;
                    }
                    if constexpr (RGB) {
#pragma unroll
                        for (int r = 0; r < 3; ++r)
                            rgbp[((tm * 2 + rw) * 8 + q) * 3 + r] = fmaf(v, rw3[r], rgbp[((tm * 2 + rw) * 8 + q) * 3 + r]);
                    }
                }
            }
    }
    if (p.fused && p.y_absmax) publish_absmax(p.y_absmax, vmax, blockIdx.x * 4 + wave);
    if constexpr (RGB) {
        if (do_rgb) {
            // reduce-scatter over the 32 channel lanes: at the step with lane bit m the lane keeps one half of its values
            // and adds the partner's copy of that half; after 5 steps lane l31 owns the 3 sums of position l31
            // (position index = (tm*2 + rw)*8 + q, exactly the order of rgbp): 93 exchanges instead of 5 x 96
            int n = TMW * 16 * 3;
#pragma unroll
            for (int m = 16; m >= 1; m >>= 1) {
                n >>= 1;
                const bool up = (l31 & m) != 0;
#pragma unroll
                for (int i = 0; i < TMW * 16 * 3 / 2; ++i) {
                    if (i < n) {
                        const float keep = up ? rgbp[i + n] : rgbp[i];
                        const float give = up ? rgbp[i] : rgbp[i + n];
                        rgbp[i] = keep + __shfl_xor(give, m);
                    }
                }
            }
            const int pos = l31;                                  // (tm*2 + rw)*8 + q
            const int tm = pos >> 4, rw = (pos >> 3) & 1, q = pos & 7;
            const int m = m0 + 2 * (wm * TMW + tm) + rw, nn = n0 + 8 * (q >> 2) + 4 * h + (q & 3);
            if (m < ph.mh && nn < ph.mw) {
                const int part = tn_blk * WN + wn;
                float4* dst = reinterpret_cast<float4*>(p.rgb_part) +
                              (((size_t)part * p.B + b) * p.Ho + (ph.sy * m + ph.oy0)) * p.Wo + (ph.sx * nn + ph.ox0);
                *dst = make_float4(rgbp[0], rgbp[1], rgbp[2], 0.f);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The stride-2 transposed 3x3 convolution (mode CONVT3X3_UP2) with its four output phases MERGED in one block:
// the input patch of a K chunk is staged once and its 4+2+2+1 = 9 taps feed four accumulator sets (one per output
// parity), instead of four passes that each re-stage the patch for 4, 2, 2 and 1 taps.  M = 8x16 input positions,
// N = 64 output channels per block (4 phases x 2 x 1 MFMA tiles per wave = 128 accumulator registers).
// The taps are grouped by their LDS shift (dy, dx) so that each shifted A fragment is read once per chunk:
//   shift ( 0, 0): phase 0 w[0], phase 1 w[1], phase 2 w[3], phase 3 w[4]
//   shift (-1, 0): phase 0 w[6], phase 1 w[7]        shift (0,-1): phase 0 w[2], phase 2 w[5]
//   shift (-1,-1): phase 0 w[8]                       (w[k] = tap k of the 3x3 kernel, y_t[2i+ti][2j+tj] += x[i][j] w[ti][tj])
// NW = 4 waves: N = 64 channels per block, one wave per SIMD.  NW = 8 waves (2 x 4 wave grid, 512 threads): N = 128
// channels per block and two waves per SIMD (256 registers each) — the patch is staged once for twice the MFMA
// work and the second wave of a SIMD covers the barrier / staging bubbles of the first; used when the layer
// still fills the chip with the larger tile (make_plan).
#ifndef HFAGP_UP4_OCC
#define HFAGP_UP4_OCC 2
#endif
template <int KD, int NW, int IO = 0>
__global__ void __launch_bounds__(NW * 64, (NW == 4 && KD != 3) ? HFAGP_UP4_OCC : 1) upconv_bf16_kernel(const ConvParams p) {
    constexpr int NP = kind_parts_a(KD), NPB = kind_parts(KD);    // parts of the activations (LDS patch) / of the weight image
    constexpr bool F16 = kind_f16(KD);
    constexpr bool XH = (IO & 1) != 0, YH = (IO & 2) != 0;       // fp16 storage of x / y_t (see modconv_bf16_kernel)
    constexpr int XB = XH ? 2 : 4;
    static_assert(IO == 0 || KD == 1, "fp16 storage goes with the single-pass fp16 arithmetic");
    constexpr int NTH = NW * 64;
    constexpr int TM = 2, TN = 1, WN = NW / 2, BM = 128, BNU = WN * TN * 32, PH = BM / PW, NITEM = 9, RB = kind_nprod(KD) == 1 ? 6 : 3;
    constexpr int LPWB = RowPitch<NP>::value;
    constexpr int APOS = (PH + 2) * LPWB;
    constexpr int A_PART = APOS * APITCH, A_BUF = NP * A_PART;
    constexpr int I_GRP[NITEM] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
    constexpr int I_PHASE[NITEM] = {0, 1, 2, 3, 0, 1, 0, 2, 0};
    constexpr int I_W[NITEM] = {0, 1, 3, 4, 6, 7, 2, 5, 8};
    constexpr int G_FIRST[4] = {0, 4, 6, 8};               // first item of each shift group
    constexpr int G_OFF[4] = {(1 * LPWB + 1) * APITCH, (0 * LPWB + 1) * APITCH, (1 * LPWB + 0) * APITCH, 0};
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    char* As = lds_raw;
    float* Ss = reinterpret_cast<float*>(lds_raw + 2 * A_BUF);

    // ---- tiling (round 6).  The position grid of the even parity is (H+1) x (W+1): tiling it per sample in 8 x 16 tiles rounds BOTH
    // extents up (257 -> 33 x 17 tiles instead of 32 x 16: 1.10 x the MFMA work at 256^2, 1.41 x at 64^2, 1.88 x at 32^2).  Now:
    //   * the rows of all samples are STACKED with pitch RP = H+1 (row r = b RP + m; row m = H of a sample is the zero padding
    //     below its image, which is also what the tap (-1, .) of row 0 of the next sample has to see): B (H+1) rows tile in 8s
    //     without a remainder per sample; a tile may straddle samples (styles / range-guard scales of up to `up_ns` samples);
    //   * the columns 0 .. W-1 are tiled in 16s exactly (W % 16 == 0) and the one remaining column n = W (only x[.][W-1] reaches
    //     it: taps (0,-1), (-1,-1)) goes to FRINGE tiles: the same 8 x 16 tile and K loop, but tile column j stands for
    //     (row block j >> 1, image column W-1 + (j & 1)) — eight two-column pieces of eight different 8-row blocks, of which the odd
    //     columns are stored.  64 useful positions per fringe tile; B (H+1) / 64 of them per layer (0.8 % of the tiles at 256^2).
    unsigned id = p.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int tiles_nu = p.Cout / BNU;
    const int tn_blk = __builtin_amdgcn_readfirstlane(id % tiles_nu);  id /= tiles_nu;
    const int n_reg = p.up_tr * p.up_tw, n_tile = n_reg + p.up_nf;
    const int T = __builtin_amdgcn_readfirstlane(id % n_tile);         id /= n_tile;
    const int ks = __builtin_amdgcn_readfirstlane(id);
    const bool fringe = T >= n_reg;                                    // block-uniform
    const int RP = p.up_rp, R_total = p.up_rows;
    // regular tile: rows r0 .. r0+7, columns n0 .. n0+15; fringe tile: rows r0 .. r0+63
    const int r0 = fringe ? (T - n_reg) * 64 : (T / p.up_tw) * PH, n0 = fringe ? 0 : (T % p.up_tw) * PW;
    const int co0 = tn_blk * BNU;
    const int b_lo = min(max(r0 - 1, 0) / RP, p.B - 1);                // first sample the patch touches

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5, l31 = lane & 31;
    const int c_begin = __builtin_amdgcn_readfirstlane((int)(((long long)p.nchunks * ks) / p.ksplit));
    const int c_end = __builtin_amdgcn_readfirstlane((int)(((long long)p.nchunks * (ks + 1)) / p.ksplit));

    // ---- A staging (as in modconv_bf16_kernel): patch rows r0-1 .. r0+PH-1, columns n0-1 .. n0+PW-1 (fringe: see above)
    const int npatch = p.ph * p.pw;
    constexpr int A_PER_T = ((PH + 2) * (PW + 2) * 4 + NTH - 1) / NTH;
    static_assert(A_PER_T == 2 || A_PER_T == 3, "staging schedule: two or three slots per thread");
    float4 ra[A_PER_T];
    const char* xb = reinterpret_cast<const char*>(p.x) + (long long)b_lo * p.x_batch_stride * XB;
    // styles of the up_ns samples from b_lo on (ones past the batch), then the fp16 range-guard scales 2^-e | 2^e per sample
    float* Gd = Ss + p.up_ns * p.Cin;                              // [up_ns] 2^-e, then [up_ns] 2^e
    {
        const long long s_lo = (long long)b_lo * p.Cin, s_n = (long long)p.B * p.Cin;
        for (int i = tid; i < p.up_ns * p.Cin; i += NTH) Ss[i] = (p.styles && s_lo + i < s_n) ? p.styles[s_lo + i] : 1.f;
        for (int sI = 0; sI < p.up_ns; ++sI) {
            float bk = 1.f, dn = 1.f;
            if constexpr (F16)
                dn = style_range_guard((p.styles && b_lo + sI < p.B) ? p.styles + (size_t)(b_lo + sI) * p.Cin : nullptr, p.Cin, lane, &bk,
                                       p.x_absmax);
            if (tid == 0) { Gd[sI] = dn; Gd[p.up_ns + sI] = bk; }
        }
    }
    __syncthreads();
    unsigned aoff[A_PER_T];
    int lds_a[A_PER_T], soff[A_PER_T];
    float amask[A_PER_T];
#pragma unroll
    for (int k = 0; k < A_PER_T; ++k) {
        const int idx = min(tid + k * NTH, npatch * 4 - 1);
        const int pix = idx >> 2, q = idx & 3;
        const int pi = pix / p.pw, pj = pix % p.pw;
        lds_a[k] = (pi * LPWB + pj) * APITCH + 8 * q;
        int r, n;
        if (!fringe) { r = r0 - 1 + pi; n = n0 - 1 + pj; }
        else { r = pj == 0 ? -1 : r0 + 8 * ((pj - 1) >> 1) + pi - 1; n = p.W - 1 + ((pj - 1) & 1); }
        const bool row_ok = r >= 0 && r < R_total;
        const int bb = row_ok ? r / RP : b_lo, m = r - bb * RP;
        const bool inside = row_ok && m < p.H && n >= 0 && n < p.W;
        const int sel = inside ? bb - b_lo : 0;
        aoff[k] = inside ? (unsigned)(((long long)sel * p.x_batch_stride + (long long)(m * p.W + n) * p.Cin + 4 * q) * XB) : 0u;
        amask[k] = inside ? Gd[sel] : 0.f;          // zero padding and the fp16 range guard in one factor
        soff[k] = sel * p.Cin + 4 * q;
    }
    auto load_a = [&](int chunk) __attribute__((always_inline)) {
        const char* xc = xb + (long long)chunk * (CKB * XB);
#pragma unroll
        for (int k = 0; k < A_PER_T; ++k) {
            if constexpr (XH) {
                const uint2 u = *reinterpret_cast<const uint2*>(xc + aoff[k]);
                ra[k].x = __builtin_bit_cast(float, u.x);
                ra[k].y = __builtin_bit_cast(float, u.y);
            } else {
                ra[k] = *reinterpret_cast<const float4*>(xc + aoff[k]);
            }
        }
    };
    auto store_a = [&](int chunk, auto buf_tag, auto k_tag) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_tag)::value, k = decltype(k_tag)::value;
        {
            const float m = amask[k];
            const float4 sv = *reinterpret_cast<const float4*>(Ss + chunk * CKB + soff[k]);
            uint2 parts[NP];
            if constexpr (XH) {
                const f32x2 s01 = {sv.x * m, sv.y * m}, s23 = {sv.z * m, sv.w * m};
                const f16x2 x01 = __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, ra[k].x));
                const f16x2 x23 = __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, ra[k].y));
                parts[0] = make_uint2(__builtin_bit_cast(unsigned, x01 * __builtin_convertvector(s01, f16x2)),
                                      __builtin_bit_cast(unsigned, x23 * __builtin_convertvector(s23, f16x2)));
            } else
            split4<KD>(make_float4(ra[k].x * (sv.x * m), ra[k].y * (sv.y * m), ra[k].z * (sv.z * m),
                                   ra[k].w * (sv.w * m)), parts);     // (F16X2: one saturating fp16 part)
#pragma unroll
            for (int q = 0; q < NP; ++q)
                *reinterpret_cast<uint2*>(As + BUF * A_BUF + q * A_PART + lds_a[k]) = parts[q];
        }
    };

    // ---- B fragments: one 32-column tile per wave, ring of RB items
    const char* wb = reinterpret_cast<const char*>(p.wt);
    const int cq8 = p.Cin >> 3;
    const int part_stride = 9 * cq8 * p.Cout;
    unsigned bth[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bth[tn] = (unsigned)(h * p.Cout + co0 + (wn * TN + tn) * 32 + l31) * 16u;
    int apos[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int pidx = (wm * TM + tm) * 32 + l31;
        apos[tm] = ((pidx >> 4) * LPWB + (pidx & 15)) * APITCH + 16 * h;
    }

    f32x16 acc[4][TM][TN];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[f][tm][tn][r] = 0.f;

    constexpr int NPROD = kind_nprod(KD);
    constexpr int PA[6] = {kind_pa(KD, 0), kind_pa(KD, 1), kind_pa(KD, 2), kind_pa(KD, 3), kind_pa(KD, 4), kind_pa(KD, 5)};
    constexpr int PB[6] = {kind_pb(KD, 0), kind_pb(KD, 1), kind_pb(KD, 2), kind_pb(KD, 3), kind_pb(KD, 4), kind_pb(KD, 5)};
    u32x4 bq[RB][TN][NPB];
    u32x4 af[2][TM][NP];                  // A fragments of the current and the next shift group
    auto issue_b = [&](int c, auto i_tag, auto slot_tag) __attribute__((always_inline)) {
        constexpr int I = decltype(i_tag)::value, SL = decltype(slot_tag)::value;
        const int cc = min(c, c_end - 1);
#pragma unroll
        for (int q = 0; q < NPB; ++q) {
            const char* base = wb + (long long)(q * part_stride + (I_W[I] * cq8 + cc * 2) * p.Cout) * 16;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bq[SL][tn][q] = *reinterpret_cast<const u32x4*>(base + bth[tn]);
        }
    };
    auto read_a = [&](auto u_tag, auto g_tag) __attribute__((always_inline)) {
        constexpr int UU = decltype(u_tag)::value, G = decltype(g_tag)::value;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int q = 0; q < NP; ++q)
                af[G & 1][tm][q] = *reinterpret_cast<const u32x4*>(As + UU * A_BUF + q * A_PART + G_OFF[G] + apos[tm]);
    };
    auto item = [&](int c, auto u_tag, auto i_tag) __attribute__((always_inline)) {
        constexpr int I = decltype(i_tag)::value, G = I_GRP[I], F = I_PHASE[I];
        constexpr int SL = (decltype(u_tag)::value * NITEM + I) % RB;     // ring slot: repeats every chunk pair
        if constexpr (I == G_FIRST[G] && G < 3) read_a(u_tag, std::integral_constant<int, G + 1>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pr = 0; pr < NPROD; ++pr)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[F][tm][tn] = mfma16<F16>(af[G & 1][tm][PA[pr]], bq[SL][tn][PB[pr]], acc[F][tm][tn]);
        // the patch of the next chunk is converted into the other LDS buffer under the last A_PER_T items
        if constexpr (I >= NITEM - A_PER_T)
            store_a(min(c + 1, c_end - 1), std::integral_constant<int, 1 - decltype(u_tag)::value>{},
                    std::integral_constant<int, I - (NITEM - A_PER_T)>{});
        issue_b(c + (I + RB) / NITEM, std::integral_constant<int, (I + RB) % NITEM>{}, std::integral_constant<int, SL>{});
        if constexpr (I == NITEM - 1) load_a(min(c + 2, c_end - 1));
        __builtin_amdgcn_sched_barrier(0);
    };
    auto chunk = [&](int c, auto u_tag) __attribute__((always_inline)) {
        __syncthreads();                                    // publishes the patch of chunk c
        read_a(u_tag, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        item(c, u_tag, std::integral_constant<int, 0>{});
        item(c, u_tag, std::integral_constant<int, 1>{});
        item(c, u_tag, std::integral_constant<int, 2>{});
        item(c, u_tag, std::integral_constant<int, 3>{});
        item(c, u_tag, std::integral_constant<int, 4>{});
        item(c, u_tag, std::integral_constant<int, 5>{});
        item(c, u_tag, std::integral_constant<int, 6>{});
        item(c, u_tag, std::integral_constant<int, 7>{});
        item(c, u_tag, std::integral_constant<int, 8>{});
    };
    static_assert((2 * NITEM) % RB == 0 && RB <= NITEM, "ring slots must repeat every chunk pair");
    if (c_begin < c_end) {
        __syncthreads();                                    // styles are in LDS
        load_a(c_begin);
        issue_b(c_begin, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        issue_b(c_begin, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        issue_b(c_begin, std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
        if constexpr (RB > 3) {
            issue_b(c_begin, std::integral_constant<int, 3>{}, std::integral_constant<int, 3>{});
            issue_b(c_begin, std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});
            issue_b(c_begin, std::integral_constant<int, 5>{}, std::integral_constant<int, 5>{});
        }
        store_a(c_begin, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        store_a(c_begin, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
        if constexpr (A_PER_T > 2) store_a(c_begin, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
        load_a(min(c_begin + 1, c_end - 1));
        // pairs of chunks (LDS buffer = chunk parity = compile time), then the odd one: an exit in the MIDDLE of the
        // loop body made the register allocator keep the 128 accumulators in two AGPR sets and copy them
        // (~190 v_accvgpr_mov per 108 MFMAs)
        int cg = c_begin;
        for (; cg + 1 < c_end; cg += 2) {
            chunk(cg, std::integral_constant<int, 0>{});
            chunk(cg + 1, std::integral_constant<int, 1>{});
        }
        if (cg < c_end) chunk(cg, std::integral_constant<int, 0>{});
    }

    // ---- raw stores of the four phases: y_t[2m + (f>>1)][2n + (f&1)], extents (H+1-(f>>1)) x (W+1-(f&1));
    // one 64-bit row pointer per (phase, tile, patch row), 32-bit column offsets
    float* out = p.out + (size_t)ks * p.slab;
    if (!fringe) {
        // the four patch rows of this wave (wave-uniform): sample, image row, range-guard scale
        int rb[2 * TM], rm[2 * TM];
        float rs[2 * TM];
#pragma unroll
        for (int i = 0; i < 2 * TM; ++i) {
            const int r = r0 + 2 * (wm * TM + (i >> 1)) + (i & 1);
            const bool ok = r < R_total;
            rb[i] = ok ? r / RP : 0;
            rm[i] = ok ? r - rb[i] * RP : (1 << 30);
            rs[i] = F16 ? Gd[p.up_ns + min(max(rb[i] - b_lo, 0), p.up_ns - 1)] : 1.f;
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int mh = p.H + 1 - (f >> 1), mw = p.W + 1 - (f & 1);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int co = co0 + (wn * TN + tn) * 32 + l31;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int rw = 0; rw < 2; ++rw) {
                        const int m = rm[2 * tm + rw];
                        if (m >= mh) continue;
                        const size_t rowoff = (((size_t)rb[2 * tm + rw] * p.Ho + 2 * m + (f >> 1)) * p.Wo + (f & 1)) * p.Cout + co;
                        float* rowp = out + rowoff;
                        _Float16* rowh = reinterpret_cast<_Float16*>(p.out) + rowoff;
                        const float sb = rs[2 * tm + rw];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const int n = n0 + 8 * (q >> 2) + 4 * h + (q & 3);
                            if (n >= mw) continue;
                            {
                                const float v = F16 ? acc[f][tm][tn][8 * rw + q] * sb : acc[f][tm][tn][8 * rw + q];
                                if constexpr (YH) rowh[2 * n * p.Cout] = (_Float16)v;
                                else rowp[2 * n * p.Cout] = v;
                            }
                        }
                    }
            }
        }
    } else {
        // fringe tile: tile column j = 8 (q >> 2) + 4 h + (q & 3) is odd exactly for odd q; it is image column W of row
        // r0 + 8 (j >> 1) + patch row -> y_t[2m + fy][2W] of the two even-column parities (f = 0, 2)
#pragma unroll
        for (int f = 0; f < 4; f += 2) {
            const int mh = p.H + 1 - (f >> 1);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int co = co0 + (wn * TN + tn) * 32 + l31;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int rw = 0; rw < 2; ++rw)
#pragma unroll
                        for (int q = 1; q < 8; q += 2) {
                            const int j = 8 * (q >> 2) + 4 * h + (q & 3);
                            const int r = r0 + 8 * (j >> 1) + 2 * (wm * TM + tm) + rw;
                            if (r >= R_total) continue;
                            const int bb = r / RP, m = r - bb * RP;
                            if (m >= mh) continue;
                            const size_t off = (((size_t)bb * p.Ho + 2 * m + (f >> 1)) * p.Wo + 2 * p.W) * p.Cout + co;
                            const float v = F16 ? acc[f][tm][tn][8 * rw + q] * Gd[p.up_ns + min(bb - b_lo, p.up_ns - 1)]
                                                : acc[f][tm][tn][8 * rw + q];
                            if constexpr (YH) reinterpret_cast<_Float16*>(p.out)[off] = (_Float16)v;
                            else out[off] = v;
                        }
            }
        }
    }
}

template <int NP, int TM>
static size_t bf16_lds_bytes(int cin) {
    constexpr int PH = 2 * TM * 32 / PW;
    return (size_t)2 * NP * (PH + 2) * RowPitch<NP>::value * APITCH + (size_t)(cin + 8) * sizeof(float);   // + guard scratch
}

template <int KD>
static void launch_group(const Plan& pl, int phase0, int nphase, int ntaps, int cin, hipStream_t s) {
    const dim3 grid(pl.grid.x, (unsigned)nphase, 1);
    const size_t lds = bf16_lds_bytes<kind_parts_a(KD), 2>(cin);
    switch (ntaps) {
        case 9: modconv_bf16_kernel<KD, 2, 9><<<grid, 256, lds, s>>>(pl.p, phase0); break;
        case 4: modconv_bf16_kernel<KD, 2, 4><<<grid, 256, lds, s>>>(pl.p, phase0); break;
        case 2: modconv_bf16_kernel<KD, 2, 2><<<grid, 256, lds, s>>>(pl.p, phase0); break;
        default: modconv_bf16_kernel<KD, 2, 1><<<grid, 256, lds, s>>>(pl.p, phase0); break;
    }
}

// mode CONVS2_BWD, the four parity phases merged in one block (make_plan: merged_s2; grid.y = 1)
template <int KD>
static void launch_s2_merged(const Plan& pl, int cin, hipStream_t s) {
    modconv_bf16_kernel<KD, 2, 0><<<pl.grid, 256, bf16_lds_bytes<kind_parts_a(KD), 2>(cin), s>>>(pl.p, 0);
}

// LDS of the merged up-conv: the two patch buffers + styles and range-guard scales of up_ns samples; beyond the 64 KB default the
// kernel's dynamic-LDS limit is raised once per instantiation
template <int KD, int NW, int IO>
static void launch_up_one(const Plan& pl, int cin, hipStream_t s) {
    const size_t lds = bf16_lds_bytes<kind_parts_a(KD), 2>(0) + (size_t)(pl.p.up_ns * (cin + 2)) * sizeof(float);
    static bool raised = false;
    if (lds > 64 * 1024 && !raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&upconv_bf16_kernel<KD, NW, IO>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        raised = true;
    }
    upconv_bf16_kernel<KD, NW, IO><<<pl.grid, NW * 64, lds, s>>>(pl.p);
}

// fp16-storage variants (KD = 1 only): io = x_f16 | y_f16 << 1
static void launch_up_io(const Plan& pl, int cin, int io, hipStream_t s) {
    const bool w8 = pl.up_waves == 8;
    if (io == 2) {
        if (w8) launch_up_one<1, 8, 2>(pl, cin, s); else launch_up_one<1, 4, 2>(pl, cin, s);
    } else {
        if (w8) launch_up_one<1, 8, 3>(pl, cin, s); else launch_up_one<1, 4, 3>(pl, cin, s);
    }
}

template <int KD>
static void launch_up(const Plan& pl, int cin, hipStream_t s) {
    if constexpr (KD != 3) {            // (three parts: the 8-wave variant would spill; make_plan never asks for it)
        if (pl.up_waves == 8) {
            launch_up_one<KD, 8, 0>(pl, cin, s);
            return;
        }
    }
    launch_up_one<KD, 4, 0>(pl, cin, s);
}

int launch_modconv_bf16(const HfagpModconvArgs* a, Plan& pl, hipStream_t s) {
    HFAGP_REQUIRE(a->Cin % CKB == 0 && (a->Cout % BNB == 0 || (a->Cout % BNB >= 96 && !pl.merged_up)), HFAGP_EUNSUPPORTED,
                  "modconv (16-bit MFMA): Cin=%d must be a multiple of %d and Cout=%d of %d (or 96 mod 128, 512-B tail pad)",
                  a->Cin, CKB, a->Cout, BNB);
    HFAGP_REQUIRE(pl.bn == BNB && pl.bm == 128, HFAGP_EUNSUPPORTED, "modconv (16-bit MFMA): unexpected plan");
    HFAGP_REQUIRE(a->Cin <= 512, HFAGP_EUNSUPPORTED, "modconv (16-bit MFMA): Cin=%d > 512 (style image in LDS)", a->Cin);
    const int kd = kind_of(a->precision);
    HFAGP_REQUIRE(kd != 0, HFAGP_EBADARG, "modconv: unknown precision %d", a->precision);
    const int io = (a->x_f16 ? 1 : 0) | (a->y_f16 ? 2 : 0);
    if (io) {
        const ConvParams& q = pl.p;
        HFAGP_REQUIRE(kd == 1 && q.ksplit * q.nslab == 1 && a->Cout % BNB == 0 &&
                          ((a->mode == HFAGP_CONV3X3 && io == 3) || (a->mode == HFAGP_CONVT3X3_UP2 && (io & 2))),
                      HFAGP_EUNSUPPORTED, "modconv: fp16 storage needs precision F16, no split-K, Cout %% 128 == 0 and mode 0 "
                                          "(x and y fp16) or mode 1 (y fp16); got precision %d mode %d x_f16 %d y_f16 %d ksplit %d",
                      a->precision, a->mode, a->x_f16, a->y_f16, q.ksplit);
        if (a->mode == HFAGP_CONVT3X3_UP2) {
            HFAGP_REQUIRE(pl.merged_up, HFAGP_EUNSUPPORTED, "modconv: fp16 storage of the up-conv needs the merged four-phase "
                                                            "kernel (grids >= 32x32 at Cin <= 512)");
            HFAGP_REQUIRE(q.up_ns <= 1 || (long long)q.up_ns * a->x_batch_stride * (a->x_f16 ? 2 : 4) < (1ll << 32), HFAGP_EUNSUPPORTED,
                          "modconv (merged up-conv): %d samples of %lld elements exceed the 32-bit patch offsets", q.up_ns,
                          (long long)a->x_batch_stride);
            launch_up_io(pl, a->Cin, io, s);
            return check_launch("modconv_fwd (fp16 storage, merged up-conv)");
        }
        const dim3 grid(pl.grid.x, 1, 1);
        modconv_bf16_kernel<1, 2, 9, 3><<<grid, 256, bf16_lds_bytes<1, 2>(a->Cin), s>>>(pl.p, 0);
        return check_launch("modconv_fwd (fp16 storage)");
    }
    // the kernel is specialised on the tap count: one launch per run of phases with the same number of taps
    // (3x3: one; stride-2 transposed conv and its adjoint: 4 | 2, 2 | 1)
    const ConvParams& p = pl.p;
    if (pl.merged_up) {                 // one block for the four phases (grid.y = 1)
        // (a block addresses the up_ns samples its tile can touch with 32-bit byte offsets from the first one)
        HFAGP_REQUIRE(p.up_ns <= 1 || (long long)p.up_ns * a->x_batch_stride * (a->x_f16 ? 2 : 4) < (1ll << 32), HFAGP_EUNSUPPORTED,
                      "modconv (merged up-conv): %d samples of %lld elements exceed the 32-bit patch offsets", p.up_ns,
                      (long long)a->x_batch_stride);
        switch (kd) {
            case 1: launch_up<1>(pl, a->Cin, s); break;
            case 2: launch_up<2>(pl, a->Cin, s); break;
            case 3: launch_up<3>(pl, a->Cin, s); break;
            case 5: launch_up<5>(pl, a->Cin, s); break;
            default: launch_up<4>(pl, a->Cin, s); break;
        }
        return check_launch("modconv_fwd (16-bit MFMA, merged up-conv)");
    }
    if (pl.merged_s2) {
        switch (kd) {
            case 1: launch_s2_merged<1>(pl, a->Cin, s); break;
            case 2: launch_s2_merged<2>(pl, a->Cin, s); break;
            case 3: launch_s2_merged<3>(pl, a->Cin, s); break;
            case 5: launch_s2_merged<5>(pl, a->Cin, s); break;
            default: launch_s2_merged<4>(pl, a->Cin, s); break;
        }
        return check_launch("modconv_fwd (16-bit MFMA, merged adjoint of the up-conv)");
    }
    for (int p0 = 0; p0 < p.nphase;) {
        int n = 1;
        while (p0 + n < p.nphase && p.phase[p0 + n].ntaps == p.phase[p0].ntaps) ++n;
        const int nt = p.phase[p0].ntaps;
        HFAGP_REQUIRE(nt == 9 || nt == 4 || nt == 2 || nt == 1, HFAGP_EUNSUPPORTED, "modconv (16-bit MFMA): %d taps", nt);
        switch (kd) {
            case 1: launch_group<1>(pl, p0, n, nt, a->Cin, s); break;
            case 2: launch_group<2>(pl, p0, n, nt, a->Cin, s); break;
            case 3: launch_group<3>(pl, p0, n, nt, a->Cin, s); break;
            case 5: launch_group<5>(pl, p0, n, nt, a->Cin, s); break;
            default: launch_group<4>(pl, p0, n, nt, a->Cin, s); break;
        }
        p0 += n;
    }
    return check_launch("modconv_fwd (16-bit MFMA)");
}

// weight [Cout][Cin][taps] -> wb [parts][taps][Cin/8][Cout][8] bf16 or fp16 (operand kind kd); thread = (tap, ci group, co)
__global__ void __launch_bounds__(256) weight_prep_split_kernel(const float* __restrict__ w, uint4* __restrict__ wb,
                                                                int Cout, int Cin, int taps, int kd) {
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
    if (kd == 5) kd = 4;          // F16X2 reads the F16X3 image (two fp16 parts of the weights)
    if (kd == 1 || kd == 4) {
        unsigned u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = pack_f16(r[2 * e], r[2 * e + 1]);
        wb[idx] = make_uint4(u[0], u[1], u[2], u[3]);
        if (kd == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                u[e] = pack_f16(r[2 * e] - f16_lo_back(u[e]), r[2 * e + 1] - f16_hi_back(u[e]));
            wb[n + idx] = make_uint4(u[0], u[1], u[2], u[3]);
        }
        return;
    }
    const int nparts = kd;
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

// Batched weight preparation (round 5; ABI 11): while the generator is being TUNED its weights change every step, and every step
// needs, per conv layer, the forward B-operand image, the image of the Cin/Cout TRANSPOSE for the bwd-data GEMM and wsq for the
// demodulation — 47 weight_prep_split launches + 23 weight_prep launches (which also wrote an fp32 image nobody read) = 1.2 ms of a
// 15 ms step.  Here ONE launch serves all layers: a block stages a 32 (co) x 32 (ci) x taps tile of one weight in LDS with coalesced
// reads and emits the three outputs from it — the weight is read once, every output leaves in 512-byte runs.
constexpr int kWPMax = 48;
struct WPItem { const float* w; uint4* img; uint4* img_t; float* wsq; int Cout, Cin, taps, kd, kd_t, tile0; };
struct WPBatch { WPItem it[kWPMax]; int n; };

__device__ __forceinline__ void wp_emit(const float (&r)[8], uint4* dst, long long n_img, long long idx, int kd) {
    if (kd == 5) kd = 4;
    if (kd == 1 || kd == 4) {
        unsigned u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = pack_f16(r[2 * e], r[2 * e + 1]);
        dst[idx] = make_uint4(u[0], u[1], u[2], u[3]);
        if (kd == 4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = pack_f16(r[2 * e] - f16_lo_back(u[e]), r[2 * e + 1] - f16_hi_back(u[e]));
            dst[n_img + idx] = make_uint4(u[0], u[1], u[2], u[3]);
        }
        return;
    }
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = r[e];
    for (int q = 0; q < kd; ++q) {
        unsigned u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            u[e] = pack_bf16(t[2 * e], t[2 * e + 1]);
            t[2 * e] -= __builtin_bit_cast(float, u[e] << 16);
            t[2 * e + 1] -= __builtin_bit_cast(float, u[e] & 0xffff0000u);
        }
        dst[q * n_img + idx] = make_uint4(u[0], u[1], u[2], u[3]);
    }
}

__global__ void __launch_bounds__(256) weight_prep_batch_kernel(const WPBatch b) {
    __shared__ float tile[32][32 * 9 + 1];                     // [co][ci * taps + t]
    int i = 0;
    while (i + 1 < b.n && (int)blockIdx.x >= b.it[i + 1].tile0) ++i;
    const WPItem& a = b.it[i];
    const int tci_n = a.Cin >> 5;
    const int tl = blockIdx.x - a.tile0, co0 = (tl / tci_n) * 32, ci0 = (tl % tci_n) * 32;
    const int taps = a.taps, row = 32 * taps;
    for (int e = threadIdx.x; e < 32 * row; e += 256) {
        const int co = e / row, r = e - co * row;
        tile[co][r] = a.w[((size_t)(co0 + co) * a.Cin + ci0) * taps + r];
    }
    __syncthreads();
    const long long n_f = (long long)taps * (a.Cin >> 3) * a.Cout, n_t = (long long)taps * (a.Cout >> 3) * a.Cin;
    for (int e = threadIdx.x; e < taps * 4 * 32; e += 256) {
        const int c = e & 31, g = (e >> 5) & 3, t = e >> 7;
        float r[8];
        if (a.img) {                                           // forward image: 8 consecutive ci of (tap t, co c)
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] = tile[c][(8 * g + k) * taps + t];
            wp_emit(r, a.img, n_f, ((long long)t * (a.Cin >> 3) + (ci0 >> 3) + g) * a.Cout + co0 + c, a.kd);
        }
        if (a.img_t) {                                         // image of the transpose: 8 consecutive co of (tap t, ci c)
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] = tile[8 * g + k][c * taps + t];
            wp_emit(r, a.img_t, n_t, ((long long)t * (a.Cout >> 3) + (co0 >> 3) + g) * a.Cin + ci0 + c, a.kd_t);
        }
    }
    if (a.wsq)
        for (int e = threadIdx.x; e < 32 * 32; e += 256) {
            const int ci = e & 31, co = e >> 5;
            float sq = 0.f;
            for (int t = 0; t < taps; ++t) { const float v = tile[co][ci * taps + t]; sq += v * v; }
            a.wsq[(size_t)(co0 + co) * a.Cin + ci0 + ci] = sq;
        }
}

}  // namespace hfagp

using namespace hfagp;

static int weight_prep_kind(const float* weight, void* wb, int32_t Cout, int32_t Cin, int32_t taps, int kd, void* stream) {
    HFAGP_REQUIRE(weight && wb, HFAGP_EBADARG, "weight_prep_split: null pointer");
    HFAGP_REQUIRE(Cin % 8 == 0 && Cout > 0 && (taps == 1 || taps == 9), HFAGP_EUNSUPPORTED,
                  "weight_prep_split: Cin=%d must be a multiple of 8, taps=%d in {1,9}", Cin, taps);
    const long long n = (long long)taps * (Cin / 8) * Cout;
    weight_prep_split_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        weight, reinterpret_cast<uint4*>(wb), Cout, Cin, taps, kd);
    return check_launch("weight_prep_split");
}

extern "C" int hfagp_weight_prep_split(const float* weight, void* wb, int32_t Cout, int32_t Cin, int32_t taps,
                                       int32_t nparts, void* stream) {
    HFAGP_REQUIRE(nparts >= 1 && nparts <= 3, HFAGP_EUNSUPPORTED, "weight_prep_split: nparts=%d in {1,2,3}", nparts);
    return weight_prep_kind(weight, wb, Cout, Cin, taps, nparts, stream);
}

extern "C" int hfagp_weight_prep_prec(const float* weight, void* wb, int32_t Cout, int32_t Cin, int32_t taps,
                                      int32_t precision, void* stream) {
    const int kd = kind_of(precision);
    HFAGP_REQUIRE(kd != 0, HFAGP_EBADARG, "weight_prep_prec: precision %d has no 16-bit weight image", precision);
    return weight_prep_kind(weight, wb, Cout, Cin, taps, kd, stream);
}

extern "C" int hfagp_weight_prep_batch(const HfagpWeightPrepItem* items, int32_t n, void* stream) {
    HFAGP_REQUIRE(items && n >= 1 && n <= kWPMax, HFAGP_EBADARG, "weight_prep_batch: 1..%d items", kWPMax);
    WPBatch b;
    int tiles = 0;
    for (int i = 0; i < n; ++i) {
        const HfagpWeightPrepItem& a = items[i];
        HFAGP_REQUIRE(a.weight && (a.image || a.image_t || a.wsq), HFAGP_EBADARG, "weight_prep_batch: null pointer (item %d)", i);
        HFAGP_REQUIRE(a.Cout % 32 == 0 && a.Cin % 32 == 0 && (a.taps == 1 || a.taps == 9), HFAGP_EUNSUPPORTED,
                      "weight_prep_batch: item %d: Cout=%d, Cin=%d must be multiples of 32, taps=%d in {1,9}", i, a.Cout, a.Cin, a.taps);
        const int kd = a.image ? kind_of(a.precision) : 0, kd_t = a.image_t ? kind_of(a.precision_t) : 0;
        HFAGP_REQUIRE((!a.image || kd != 0) && (!a.image_t || kd_t != 0), HFAGP_EBADARG,
                      "weight_prep_batch: item %d: precision without a 16-bit weight image", i);
        b.it[i] = WPItem{a.weight, reinterpret_cast<uint4*>(a.image), reinterpret_cast<uint4*>(a.image_t), a.wsq, a.Cout, a.Cin, a.taps,
                         kd, kd_t, tiles};
        tiles += (a.Cout / 32) * (a.Cin / 32);
    }
    b.n = n;
    weight_prep_batch_kernel<<<(unsigned)tiles, 256, 0, (hipStream_t)stream>>>(b);
    return check_launch("weight_prep_batch");
}

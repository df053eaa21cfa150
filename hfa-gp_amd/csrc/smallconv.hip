// Modulated conv on SMALL images (round 4: VERDICT r3 #2, "small-image kernels"): the 4^2 ... 16^2 layers of the backbone (3x3,
// its data adjoint, and the 1x1 toRGB) at the batch sizes the reference trains and re-enacts at (2 and 1).
//
// There the implicit-GEMM kernel of modconv_bf16.hip is a poor fit: its 128-position x 128-channel tile, LDS staging and one
// barrier per chunk make a 20 - 27 us launch out of 0.15 - 2.4 GFLOP.  Here a block owns 32 positions x 32 output channels and
// streams both operands straight from L2 / the memory-side cache into the MFMAs:
//   * MEASURED, first version: the whole K range per block with the epilogue in the same launch and no workspace — 16 ... 128
//     blocks per layer: 24 - 29 us, because 16 CUs cannot pull a layer's 9.4 MB of weights (~25 GB/s per CU), whatever the
//     prefetch depth.  So K is split over blocks after all (smallconv_ksplit: until ~256 blocks stream the weights) and the slabs
//     go through the caller's workspace and the existing reducer; what is left of the idea is the lean block:
//     3x3 at 4^2 / 8^2 / 16^2, B = 1: 19.8 / 20.6 / 26.7 us -> 12.8 / 14.5 / 19.9 us; 1x1 (toRGB): 19 - 25 us -> 7 - 14 us.
//   * 8 waves per block split K between them (wave w takes the 16-channel chunks w, w + 8, ...; all taps of a chunk), each with one
//     32 x 32 accumulator tile; a fixed-order sum through 32 KB of LDS joins them (deterministic);
//   * nothing is staged: the A operand (32 positions x 16 channels of one tap) is read straight from L2 — 32 bytes per lane, zero
//     outside the image — scaled by the style and split into its 16-bit parts in registers, as torgb_skip_kernel does; the B operand
//     comes from the same pre-split weight image as everywhere ([part][tap][Cin/8][Cout][8]); the activation loads run two steps
//     ahead of the MFMAs, the weight loads a whole chunk ahead (they come from the memory-side cache: ~0.5 us);
//   * a layer is 16 (4^2) ... 128 (16^2) blocks per sample of 8 waves: the whole K loop of a wave is 36 steps.
// Same operand arithmetic as modconv_bf16_kernel (operand kinds, fp16 range guard, product order); the K summation order differs
// (per-wave partial sums), i.e. fp32 rounding noise.  Taken by hfagp_modconv_fwd for modes 0 / 2 / 3 on 16-bit weight images when
// H * W <= 256 (the 1x1: <= 1024; modconv_plan.h smallconv_takes); the K slices go through the caller's workspace like the
// staged kernel's (hfagp_modconv_workspace_bytes: smallconv_ksplit slabs, 0 when one block takes the whole K range).
#include "conv16_common.h"

namespace hfagp {

constexpr int kSmallWaves = 8;

template <int KD, int NT>
__global__ void __launch_bounds__(kSmallWaves * 64, 2) smallconv_kernel(const ConvParams p) {
    constexpr int NPA = kind_parts_a(KD), NPB = kind_parts(KD);
    constexpr bool F16 = kind_f16(KD);
    constexpr int NPROD = kind_nprod(KD);
    constexpr int PA[6] = {kind_pa(KD, 0), kind_pa(KD, 1), kind_pa(KD, 2), kind_pa(KD, 3), kind_pa(KD, 4), kind_pa(KD, 5)};
    constexpr int PB[6] = {kind_pb(KD, 0), kind_pb(KD, 1), kind_pb(KD, 2), kind_pb(KD, 3), kind_pb(KD, 4), kind_pb(KD, 5)};
    __shared__ __attribute__((aligned(16))) float Ss[512];
    __shared__ __attribute__((aligned(16))) float red[kSmallWaves][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    const Phase& ph = p.phase[0];
    const int HW = p.Ho * p.Wo;
    const int tiles_m = (HW + 31) >> 5, tiles_n = (p.Cout + 31) >> 5;
    unsigned id = blockIdx.x;
    const int nt_blk = __builtin_amdgcn_readfirstlane(id % tiles_n); id /= tiles_n;
    const int mt_blk = __builtin_amdgcn_readfirstlane(id % tiles_m); id /= tiles_m;
    const int b = __builtin_amdgcn_readfirstlane(id % p.B); id /= p.B;
    const int ks = __builtin_amdgcn_readfirstlane(id);          // K slice of this block (p.ksplit slices: chunks [c_lo, c_hi))
    const int c_lo = __builtin_amdgcn_readfirstlane((int)(((long long)p.nchunks * ks) / p.ksplit));
    const int c_hi = __builtin_amdgcn_readfirstlane((int)(((long long)p.nchunks * (ks + 1)) / p.ksplit));
    const int co0 = nt_blk * 32;

    for (int i = tid; i < p.Cin; i += kSmallWaves * 64) Ss[i] = p.styles ? p.styles[(size_t)b * p.Cin + i] : 1.f;
    float sback = 1.f, sdown = 1.f;
    if constexpr (F16) sdown = style_range_guard(p.styles ? p.styles + (size_t)b * p.Cin : nullptr, p.Cin, lane, &sback, p.x_absmax);

    // this lane's position of the tile and, per tap, the byte offset of its 8 channels in chunk 0 + the zero-padding mask
    const int pos = mt_blk * 32 + l31;
    const int py = pos / p.Wo, px = pos - py * p.Wo;
    unsigned aoff[NT];
    float amask[NT];
    int wtap[NT];
    const int cq8 = p.Cin >> 3;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int iy = py + ph.dy[t], ix = px + ph.dx[t];
        const bool inside = pos < HW && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
        aoff[t] = inside ? (unsigned)((iy * p.in_w + ix) * p.Cin + 8 * h) * 4u : 0u;
        amask[t] = inside ? sdown : 0.f;
        wtap[t] = ph.widx[t] * cq8 * p.Cout;
    }
    const char* xb = reinterpret_cast<const char*>(p.x) + (long long)b * p.x_batch_stride * 4;
    const char* wb = reinterpret_cast<const char*>(p.wt);
    const int part_stride = p.wtaps * cq8 * p.Cout;                                    // uint4 per part
    // (Cout = 96: three 32-channel tiles, all real — no padded tile as in the 128-wide kernel)
    const unsigned bth = (unsigned)(h * p.Cout + co0 + l31) * 16u;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // a STEP = one tap of one chunk.  Groups of G steps (G % 3 == 0) are straight-line code with a three-slot ring: the loads of
    // step k + 2 are issued before the MFMAs of step k.  NT = 9: a group is one chunk; NT = 1: three chunks (c, c + 8, c + 16).
    constexpr int G = NT == 9 ? 9 : 3;
    constexpr int CPG = NT == 9 ? 1 : 3;                      // chunks per group (of this wave's chunk sequence)
    // Two rings: the A operand (this sample's activations: small, L2-hot) two steps ahead in three slots; the B operand (the
    // weights: 9.4 MB per 512 x 512 layer, read ONCE per launch, so they come from the memory-side cache at ~0.5 us) a whole
    // group ahead — slot k of the ring is refilled with step k of the NEXT group as soon as step k has been consumed.  (With
    // both operands two steps ahead a wave waited for a weight load at every step: 36 steps x ~0.45 us = 18 us per launch.)
    float4 ra[3][2];
    u32x4 rb[G][NPB];
    const int nchunks = c_hi;                                  // (this block's chunk range ends here)
    auto load_a = [&](int c, auto t_tag, auto slot_tag) __attribute__((always_inline)) {
        constexpr int T = decltype(t_tag)::value, SL = decltype(slot_tag)::value;
        const int cc = min(c, nchunks - 1);                    // look-ahead past the end re-reads valid memory
        const char* q = xb + (size_t)cc * (CKB * 4) + aoff[T];
        ra[SL][0] = *reinterpret_cast<const float4*>(q);
        ra[SL][1] = *reinterpret_cast<const float4*>(q + 16);
    };
    auto load_b = [&](int c, auto t_tag, auto slot_tag) __attribute__((always_inline)) {
        constexpr int T = decltype(t_tag)::value, SL = decltype(slot_tag)::value;
        const int cc = min(c, nchunks - 1);
#pragma unroll
        for (int qq = 0; qq < NPB; ++qq)
            rb[SL][qq] = *reinterpret_cast<const u32x4*>(wb + (long long)(qq * part_stride + wtap[T] + cc * 2 * p.Cout) * 16 + bth);
    };
    auto compute = [&](int c, auto t_tag, auto a_tag, auto b_tag) __attribute__((always_inline)) {
        constexpr int T = decltype(t_tag)::value, SA = decltype(a_tag)::value, SB = decltype(b_tag)::value;
        if (c >= nchunks) return;                              // (wave-uniform)
        const float m = amask[T];
        const float4 s0 = *reinterpret_cast<const float4*>(Ss + c * CKB + 8 * h);
        const float4 s1 = *reinterpret_cast<const float4*>(Ss + c * CKB + 8 * h + 4);
        const float4 x0 = ra[SA][0], x1 = ra[SA][1];
        uint2 p0[NPA], p1[NPA];
        split4<KD>(make_float4(x0.x * (s0.x * m), x0.y * (s0.y * m), x0.z * (s0.z * m), x0.w * (s0.w * m)), p0);
        split4<KD>(make_float4(x1.x * (s1.x * m), x1.y * (s1.y * m), x1.z * (s1.z * m), x1.w * (s1.w * m)), p1);
        u32x4 af[NPA];
#pragma unroll
        for (int q = 0; q < NPA; ++q) af[q] = u32x4{p0[q].x, p0[q].y, p1[q].x, p1[q].y};
#pragma unroll
        for (int pr = 0; pr < NPROD; ++pr) acc = mfma16<F16>(af[PA[pr]], rb[SB][PB[pr]], acc);
    };
    // step k of the group that starts at chunk c: chunk c (NT = 9: tap k) or chunk c + 8 k (NT = 1: tap 0)
    auto chunk_of = [&](int c, int k) { return NT == 9 ? c : c + kSmallWaves * k; };
    auto prologue_b = [&](int c, auto k_tag) __attribute__((always_inline)) {
        constexpr int K = decltype(k_tag)::value;
        load_b(chunk_of(c, K), std::integral_constant<int, (NT == 9 ? K : 0)>{}, std::integral_constant<int, K>{});
    };
    __syncthreads();                                           // styles are in LDS
    {
        const int c = c_lo + wave;
        prologue_b(c, std::integral_constant<int, 0>{});
        prologue_b(c, std::integral_constant<int, 1>{});
        prologue_b(c, std::integral_constant<int, 2>{});
        if constexpr (G == 9) {
            prologue_b(c, std::integral_constant<int, 3>{});
            prologue_b(c, std::integral_constant<int, 4>{});
            prologue_b(c, std::integral_constant<int, 5>{});
            prologue_b(c, std::integral_constant<int, 6>{});
            prologue_b(c, std::integral_constant<int, 7>{});
            prologue_b(c, std::integral_constant<int, 8>{});
        }
        load_a(chunk_of(c, 0), std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        load_a(chunk_of(c, 1), std::integral_constant<int, (NT == 9 ? 1 : 0)>{}, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int c = c_lo + wave; c < nchunks; c += kSmallWaves * CPG) {
        const int cn = c + kSmallWaves * CPG;                  // first chunk of the next group
        auto step = [&](auto k_tag) __attribute__((always_inline)) {
            constexpr int K = decltype(k_tag)::value;
            constexpr int K2 = K + 2;                          // the step whose A loads go out now
            if constexpr (K2 < G) {
                load_a(chunk_of(c, K2), std::integral_constant<int, (NT == 9 ? K2 : 0)>{}, std::integral_constant<int, K2 % 3>{});
            } else {
                load_a(chunk_of(cn, K2 - G), std::integral_constant<int, (NT == 9 ? K2 - G : 0)>{}, std::integral_constant<int, K2 % 3>{});
            }
            // (sched_barrier: hipcc otherwise sinks every load to just before its first use — to save registers — which undoes
            // both rings: the listing had `global_load ... s_waitcnt vmcnt(0) ... v_mfma` at every step)
            __builtin_amdgcn_sched_barrier(0);
            compute(chunk_of(c, K), std::integral_constant<int, (NT == 9 ? K : 0)>{}, std::integral_constant<int, K % 3>{},
                    std::integral_constant<int, K>{});
            // the ring slot is free: step K of the next group
            load_b(chunk_of(cn, K), std::integral_constant<int, (NT == 9 ? K : 0)>{}, std::integral_constant<int, K>{});
            __builtin_amdgcn_sched_barrier(0);
        };
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        if constexpr (G == 9) {
            step(std::integral_constant<int, 3>{});
            step(std::integral_constant<int, 4>{});
            step(std::integral_constant<int, 5>{});
            step(std::integral_constant<int, 6>{});
            step(std::integral_constant<int, 7>{});
            step(std::integral_constant<int, 8>{});
        }
    }

    // ---- join the eight partial tiles (fixed order) and finish.  C/D layout of 32x32: column (channel) = lane & 31,
    // row (position) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    float vmax = 0.f;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = wave * 2 + rr;
        float v = red[0][r][lane];
#pragma unroll
        for (int w = 1; w < kSmallWaves; ++w) v += red[w][r][lane];
        const int opos = mt_blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int co = co0 + l31;
        if (opos < HW && co < p.Cout) {
            if (p.fused) {
                const float d = (p.dcoef ? p.dcoef[(size_t)b * p.Cout + co] : 1.f) * sback;
                const float bs = p.bias ? p.bias[co] : 0.f;
                const float nz = p.noise ? p.noise[opos] * p.noise_strength : 0.f;
                v = lrelu_gain_clamp(v * d + bs + nz, p.act, p.alpha, p.gain, p.clamp);
            } else if constexpr (F16) {
                v *= sback;
            }
            vmax = fmaxf(vmax, fabsf(v));
            p.out[(size_t)ks * p.slab + ((size_t)b * HW + opos) * p.Cout + co] = v;      // (ksplit > 1: raw partial sums, slab ks)
        }
    }
    if (p.fused && p.y_absmax) publish_absmax(p.y_absmax, vmax, blockIdx.x * kSmallWaves + wave);
}

template <int KD>
static void launch_small_kind(const ConvParams& p, unsigned grid, int ntaps, hipStream_t s) {
    if (ntaps == 9) smallconv_kernel<KD, 9><<<grid, kSmallWaves * 64, 0, s>>>(p);
    else smallconv_kernel<KD, 1><<<grid, kSmallWaves * 64, 0, s>>>(p);
}

// K slices per tile: few tiles (4^2: 16 per sample) cannot pull a layer's 9.4 MB of weights through 16 CUs fast enough (measured:
// 24 - 29 us with the whole K range per block, i.e. ~25 GB/s per CU), so K is split until ~256 blocks stream them; the slabs go
// through the caller's workspace and splitk_epilogue_kernel (modconv.hip) as for the large-tile kernels.
int smallconv_ksplit(const HfagpModconvArgs* a) {
    const long long tiles = (long long)a->B * (((long long)a->H * a->W + 31) / 32) * ((a->Cout + 31) / 32);
    const int nchunks = a->Cin / CKB;
    int ks = (int)(kNumCU / (tiles > 0 ? tiles : 1));
    const int max_ks = nchunks / kSmallWaves > 0 ? nchunks / kSmallWaves : 1;      // at least one chunk per wave
    if (ks > max_ks) ks = max_ks;
    if (ks < 1) ks = 1;
    return ks;
}

int launch_smallconv(const HfagpModconvArgs* a, Plan& pl, hipStream_t s) {
    ConvParams& p = pl.p;
    p.ksplit = smallconv_ksplit(a);
    p.fused = p.ksplit == 1;
    p.out = p.ksplit == 1 ? a->y : a->workspace;
    HFAGP_REQUIRE(p.out, HFAGP_EBADARG, "modconv (small-image kernel): %d K slices need a workspace of %zu bytes", p.ksplit,
                  (size_t)p.ksplit * p.slab * sizeof(float));
    const int HW = p.Ho * p.Wo;
    const unsigned grid = (unsigned)(a->B * ((HW + 31) / 32) * ((a->Cout + 31) / 32) * p.ksplit);
    const int nt = p.phase[0].ntaps;
    switch (kind_of(a->precision)) {
        case 1: launch_small_kind<1>(p, grid, nt, s); break;
        case 2: launch_small_kind<2>(p, grid, nt, s); break;
        case 3: launch_small_kind<3>(p, grid, nt, s); break;
        case 5: launch_small_kind<5>(p, grid, nt, s); break;
        default: launch_small_kind<4>(p, grid, nt, s); break;
    }
    return check_launch("modconv_fwd (small-image kernel)");
}

}  // namespace hfagp

// The up-sampling layer with a SHORT K (Cin = 32: the first super-resolution layer, 32 -> 256 channels, 128^2 -> 256^2) as a
// streaming kernel (round 6):
//   y = bias_act(FIR(conv_transpose2d(x * s, W, stride 2)) * d + noise)          EG3D conv2d_resample(up = 2) + bias_act
// upconv_fir_kernel (upconv_fir.hip) keeps y_t on the chip too, but hands every tile through LDS to a separate FIR phase that all
// eight waves of the block run in lockstep after the K loop: for K = 288 the K loop is a quarter of the tile's time (1.39 ms per
// 32 frames = 0.067 of the matrix pipe and 0.25 of the copy ceiling, 296 bytes of scratch per lane).  Here the raw result never
// leaves the REGISTERS:
//   * v_mfma_f32_16x16x32 with the operands SWAPPED — A = weights (16 output channels x all 32 input channels of one tap),
//     B = activations (32 channels x 16 positions of one row): D[channel][position], i.e. a lane (q = lane >> 4, p = lane & 15)
//     holds FOUR CONSECUTIVE CHANNELS 4q .. 4q+3 of position column p — for each of the step's 4 position rows and 4 output parities:
//     64 accumulator registers = the lane's 8 rows x 2 columns of y_t.  One MFMA covers the whole K of a tap (3 per tap: split operands);
//   * the VERTICAL FIR runs in the lane (a block walks DOWN a column strip in steps of 4 position rows = 8 y_t rows and carries the
//     last three y_t rows in registers), the HORIZONTAL FIR takes its neighbours from lanes p + 1, p + 2 of the same 16-lane row with
//     DPP row shifts — no LDS, no barrier, no second kernel for strip boundaries (a strip of 16 position columns yields 28 finished
//     output columns; strips overlap by two position columns, a segment of a strip starts with one MFMA-only step for its carry);
//   * demodulation, noise, bias, leaky ReLU, gain and clamp follow in place and the four channels leave as ONE 16-byte store;
//   * the weights of the wave's 16 channels (9 taps x 2 parts = 72 registers) are loaded once per block; the activation patch
//     (5 x 17 positions x 32 channels, split into its 16-bit parts) is staged in LDS once per step for the four waves = 64 channels
//     of the block, double buffered: one barrier per step;
//   * 4 waves x <= 256 registers: two or three blocks per CU, so one block's FIR / stores run under another's MFMAs.
// Same operand arithmetic (operand kinds, fp16 range guard, product order) as the kernels it replaces; the FIR sums in a different
// order ((a + d) + 3 (b + c) per axis, the 1/16 and the gain folded into the demodulation coefficient): fp32 rounding noise.
#include <type_traits>
#include "conv16_common.h"

namespace hfagp {

typedef float f32x4v __attribute__((ext_vector_type(4)));


constexpr int LCIN = 32;                       // input channels (one MFMA K)
constexpr int LPR = 4;                         // position rows per step
constexpr int LPW = 17;                        // patch columns: 16 positions + the left halo
constexpr int LSLOT = 80;                      // LDS bytes per patch position and part: 32 x 16 bit + 16 B pad (conflict-free b128 reads)
constexpr int LPART = (LPR + 1) * LPW * LSLOT; // one part image of the patch
constexpr int LUNITS = (LPR + 1) * LPW * 8;    // float4 units of a patch (8 per position)

template <bool F16>
__device__ __forceinline__ f32x4v mfma16x32(u32x4 a, u32x4 b, f32x4v c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// lane i of a 16-lane row reads lane i + N of the same row (0 past the row's end)
template <int N>
__device__ __forceinline__ float row_shl(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, true));
}

template <int KD>
__global__ void __launch_bounds__(256, 2) upfir_lean_kernel(const LeanParams p) {
    constexpr int NP = kind_parts_a(KD), NPB = kind_parts(KD);
    constexpr bool F16 = kind_f16(KD);
    constexpr int NPROD = kind_nprod(KD);
    constexpr int PA[6] = {kind_pa(KD, 0), kind_pa(KD, 1), kind_pa(KD, 2), kind_pa(KD, 3), kind_pa(KD, 4), kind_pa(KD, 5)};
    constexpr int PB[6] = {kind_pb(KD, 0), kind_pb(KD, 1), kind_pb(KD, 2), kind_pb(KD, 3), kind_pb(KD, 4), kind_pb(KD, 5)};
    static_assert(NP <= 2 && NPB <= 2, "one or two parts");
    constexpr int BUF = NP * LPART;
    // taps by shift group (as upconv_bf16_kernel): group 0 shift (0,0), 1 (-1,0), 2 (0,-1), 3 (-1,-1); phase = 2 (row parity) + column parity
    constexpr int NITEM = 9;
    constexpr int I_GRP[NITEM] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
    constexpr int I_PHASE[NITEM] = {0, 1, 2, 3, 0, 1, 0, 2, 0};
    constexpr int I_W[NITEM] = {0, 1, 3, 4, 6, 7, 2, 5, 8};
    constexpr int G_OFF[4] = {(1 * LPW + 1) * LSLOT, (0 * LPW + 1) * LSLOT, (1 * LPW + 0) * LSLOT, 0};
    __shared__ __attribute__((aligned(16))) char As[2 * BUF];
    __shared__ float Ss[LCIN];
    __shared__ __attribute__((aligned(8))) float Nz[2][8][32];     // noise of the step's 8 x 28 output pixels (staged like the patch)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, pc = lane & 15;
    unsigned id = blockIdx.x;
    const int nchb = p.Cout >> 6;
    const int chb = __builtin_amdgcn_readfirstlane(id % nchb);       id /= nchb;
    const int strip = __builtin_amdgcn_readfirstlane(id % p.nstrip); id /= p.nstrip;
    const int seg = __builtin_amdgcn_readfirstlane(id % p.nseg);     id /= p.nseg;
    const int b = __builtin_amdgcn_readfirstlane(id);
    const int s_begin = (p.nsteps * seg) / p.nseg, s_end = (p.nsteps * (seg + 1)) / p.nseg;
    const int co = chb * 64 + wave * 16;               // this wave's 16 channels
    const int n_first = 14 * strip - 1;                // image column of position column 0 (x column of patch column 1)
    const int Ho = 2 * p.H, Wo = 2 * p.W;

    // ---- styles + fp16 range guard of this sample
    if (tid < LCIN) Ss[tid] = p.styles ? p.styles[(size_t)b * LCIN + tid] : 1.f;
    float sback = 1.f, sdown = 1.f;
    if constexpr (F16) sdown = style_range_guard(p.styles ? p.styles + (size_t)b * LCIN : nullptr, LCIN, lane, &sback, p.x_absmax);

    // ---- weights of the wave's channels: A operand, lane (q, pc) = input channels 8q .. 8q+7 of output channel co + pc
    u32x4 wf[NITEM][NPB];
    {
        const u32x4* wimg = reinterpret_cast<const u32x4*>(p.wt);
        const size_t part_stride = (size_t)9 * (LCIN / 8) * p.Cout;
#pragma unroll
        for (int t = 0; t < NITEM; ++t)
#pragma unroll
            for (int pt = 0; pt < NPB; ++pt)
                wf[t][pt] = wimg[pt * part_stride + (size_t)(I_W[t] * (LCIN / 8) + q) * p.Cout + co + pc];
    }

    // ---- patch staging: unit = (patch position, 4 channels); row pi = image row 4 s - 1 + pi, column pj = image column n_first - 1 + pj
    constexpr int NU = (LUNITS + 255) / 256;
    static_assert(NU == 3, "three staging units per thread");
    int ulds[NU], urow[NU], ucol4[NU];
    float umask[NU];
    float4 ustyle[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        const int u = min(tid + k * 256, LUNITS - 1);
        const int pos = u >> 3, c4 = u & 7;
        const int pi = pos / LPW, pj = pos - pi * LPW;
        const int n = n_first - 1 + pj;
        ulds[k] = pos * LSLOT + c4 * 8;
        urow[k] = pi;
        const bool colok = n >= 0 && n < p.W;
        ucol4[k] = colok ? (n * LCIN + 4 * c4) : 0;
        umask[k] = colok ? sdown : 0.f;
        ustyle[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();                                   // styles are in LDS
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        const int c4 = min(tid + k * 256, LUNITS - 1) & 7;
        const float4 sv = *reinterpret_cast<const float4*>(Ss + 4 * c4);
        ustyle[k] = make_float4(sv.x * umask[k], sv.y * umask[k], sv.z * umask[k], sv.w * umask[k]);
    }
    const float* xb = p.x + (size_t)b * p.x_batch_stride;
    float4 ra[NU];
    auto load_patch = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            const int m = 4 * s - 1 + urow[k];
            const bool ok = m >= 0 && m < p.H;
            const float4 v = *reinterpret_cast<const float4*>(xb + (ok ? (size_t)m * p.W * LCIN + ucol4[k] : 0));
            ra[k] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_patch = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            uint2 parts[NP];
            split4<KD>(make_float4(ra[k].x * ustyle[k].x, ra[k].y * ustyle[k].y, ra[k].z * ustyle[k].z, ra[k].w * ustyle[k].w), parts);
#pragma unroll
            for (int pt = 0; pt < NP; ++pt) *reinterpret_cast<uint2*>(As + buf * BUF + pt * LPART + ulds[k]) = parts[pt];
        }
    };

    // noise tile of step s: thread = (row k = tid >> 5, column tid & 31) -> one value, fetched a step ahead like the patch
    float nzr = 0.f;
    const int nzk = tid >> 5, nzc = tid & 31;
    auto load_noise = [&](int s) __attribute__((always_inline)) {
        const int oy = 8 * s - 2 + nzk, ox = 28 * strip + nzc;
        nzr = (p.noise && nzc < 28 && oy >= 0 && oy < Ho && ox < Wo) ? p.noise[(size_t)oy * Wo + ox] : 0.f;
    };
    auto store_noise = [&](int buf) __attribute__((always_inline)) { Nz[buf][nzk][nzc] = nzr; };

    // ---- epilogue constants of the lane's four channels: out = clamp(lrelu(S dg + noise ng + bg)),  S = sum of the 16 taps with integer
    // weights: 1/16 (FIR, gain 4), the layer gain (lrelu is positively homogeneous) and the range-guard scale folded
    float dg[4], bg[4];
    {
        const int c = co + 4 * q;
        const float4 d4 = p.dcoef ? *reinterpret_cast<const float4*>(p.dcoef + (size_t)b * p.Cout + c) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 b4 = p.bias ? *reinterpret_cast<const float4*>(p.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float k = p.gain * sback * 0.0625f;
        dg[0] = d4.x * k; dg[1] = d4.y * k; dg[2] = d4.z * k; dg[3] = d4.w * k;
        bg[0] = b4.x * p.gain; bg[1] = b4.y * p.gain; bg[2] = b4.z * p.gain; bg[3] = b4.w * p.gain;
    }
    const float ng = p.noise ? p.noise_strength * p.gain : 0.f;
    const float slope = p.act == HFAGP_ACT_LRELU ? p.alpha : 1.f;
    const float cl = p.clamp >= 0.f ? p.clamp : 3.0e38f;
    const int ox0 = 28 * strip + 2 * pc;                         // the lane's two output columns ox0, ox0 + 1
    const bool col_ok = pc < 14 && ox0 < Wo;                     // (Wo is even: both or none)
    float* ycol = p.y + ((size_t)b * Ho * Wo + ox0) * p.Cout + co + 4 * q;
    const int bpos = pc * LSLOT + q * 16;                        // B-operand read offset of this lane inside a patch row

    float carry[3][2][4];                                        // y_t rows 8 s - 3 .. 8 s - 1: [row][column parity][channel]
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int fx = 0; fx < 2; ++fx)
#pragma unroll
            for (int r = 0; r < 4; ++r) carry[i][fx][r] = 0.f;
    float vmax = 0.f;

    // a segment below the top starts one step early: MFMAs only, for the three carried rows
    const int s_first = s_begin > 0 ? s_begin - 1 : 0;
    load_patch(s_first);
    load_noise(s_first);
    store_patch(s_first & 1);
    store_noise(s_first & 1);
    if (s_first + 1 < s_end) { load_patch(s_first + 1); load_noise(s_first + 1); }
#pragma unroll 1
    for (int s = s_first; s < s_end; ++s) {
        __syncthreads();                                         // patch of step s is in buffer s & 1 (and buffer (s + 1) & 1 is free)
        const char* Ab = As + (s & 1) * BUF;
        f32x4v acc[LPR][4];
#pragma unroll
        for (int pr = 0; pr < LPR; ++pr)
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[pr][f] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pr = 0; pr < LPR; ++pr) {
            u32x4 xf[4][NP];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int pt = 0; pt < NP; ++pt)
                    xf[g][pt] = *reinterpret_cast<const u32x4*>(Ab + pt * LPART + pr * LPW * LSLOT + G_OFF[g] + bpos);
            // (product outer, tap inner: consecutive MFMAs then write DIFFERENT accumulators — phases 0 1 2 3 0 1 0 2 0 — instead of
            // three dependent ones per tap)
#pragma unroll
            for (int k = 0; k < NPROD; ++k)
#pragma unroll
                for (int t = 0; t < NITEM; ++t)
                    acc[pr][I_PHASE[t]] = mfma16x32<F16>(wf[t][PB[k]], xf[I_GRP[t]][PA[k]], acc[pr][I_PHASE[t]]);
        }
        if (s + 1 < s_end) {
            store_patch((s + 1) & 1);
            store_noise((s + 1) & 1);
            if (s + 2 < s_end) { load_patch(s + 2); load_noise(s + 2); }
        }
        // ---- vertical FIR in the lane.  z[i] = y_t row 8 s - 3 + i: the three carried rows, then the step's eight (row 2 pr + fy of
        // column parity fx = accumulator phase 2 fy + fx); v[k] = (z[k] + z[k+3]) + 3 (z[k+1] + z[k+2]) is output row 8 s - 2 + k
        if (s >= s_begin) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int oy = 8 * s - 2 + k;
                float v[2][4];
#pragma unroll
                for (int fx = 0; fx < 2; ++fx)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        auto z = [&](int i) __attribute__((always_inline)) -> float {
                            return i < 3 ? carry[i < 3 ? i : 0][fx][r] : acc[i < 3 ? 0 : (i - 3) >> 1][2 * ((i + 1) & 1) + fx][r];
                        };
                        v[fx][r] = (z(k) + z(k + 3)) + 3.f * (z(k + 1) + z(k + 2));
                    }
                // ---- horizontal FIR: position column pc holds y_t columns 2 pc (fx 0) and 2 pc + 1 (fx 1) of the strip; output column
                // 2 pc takes columns 2 pc + 1 .. 2 pc + 4, output column 2 pc + 1 takes 2 pc + 2 .. 2 pc + 5
                const bool row_ok = col_ok && oy >= 0 && oy < Ho;
                const float2 nz = *reinterpret_cast<const float2*>(&Nz[s & 1][k][2 * pc]);
                float4 o0, o1;
                float e[2][4];
                const float nb0 = nz.x * ng, nb1 = nz.y * ng;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    //   h0 = v1 + 3 (v0 + v1)(p+1) + v0(p+2),   h1 = (v0 + 3 v1)(p+1) + (3 v0 + v1)(p+2)
                    // (the four row shifts stay v_mov_b32_dpp: hipcc sinks the arithmetic into the `if (row_ok)` block below, away from
                    // the moves.  Measured alternative: branch-free stores through a buffer resource, out-of-range offsets for the
                    // masked lanes — three of the four shifts then fold into v_add_f32_dpp, but the kernel ran 0.626 ms instead of 0.600)
                    const float s01 = v[0][r] + v[1][r];
                    const float t1 = fmaf(3.f, v[1][r], v[0][r]), t2 = fmaf(3.f, v[0][r], v[1][r]);
                    const float h0 = fmaf(3.f, row_shl<1>(s01), v[1][r]) + row_shl<2>(v[0][r]);
                    const float h1 = row_shl<1>(t1) + row_shl<2>(t2);
                    float y0 = fmaf(h0, dg[r], nb0 + bg[r]);
                    float y1 = fmaf(h1, dg[r], nb1 + bg[r]);
                    y0 = fmaxf(y0, y0 * slope);                 // (slope <= 1: leaky ReLU; 1: linear)
                    y1 = fmaxf(y1, y1 * slope);
                    e[0][r] = __builtin_amdgcn_fmed3f(y0, -cl, cl);
                    e[1][r] = __builtin_amdgcn_fmed3f(y1, -cl, cl);
                }
                o0 = make_float4(e[0][0], e[0][1], e[0][2], e[0][3]);
                o1 = make_float4(e[1][0], e[1][1], e[1][2], e[1][3]);
                if (row_ok) {
                    float* dst = ycol + (size_t)oy * Wo * p.Cout;
                    *reinterpret_cast<float4*>(dst) = o0;
                    *reinterpret_cast<float4*>(dst + p.Cout) = o1;
                    if (p.y_absmax)
                        vmax = fmaxf(vmax, fmaxf(fmaxf(fmaxf(fabsf(o0.x), fabsf(o0.y)), fmaxf(fabsf(o0.z), fabsf(o0.w))),
                                                 fmaxf(fmaxf(fabsf(o1.x), fabsf(o1.y)), fmaxf(fabsf(o1.z), fabsf(o1.w)))));
                }
            }
        }
        // ---- carry: the step's last three y_t rows (5, 6, 7 = position rows 2, 3, 3 with row parity 1, 0, 1)
#pragma unroll
        for (int fx = 0; fx < 2; ++fx)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                carry[0][fx][r] = acc[2][2 + fx][r];
                carry[1][fx][r] = acc[3][fx][r];
                carry[2][fx][r] = acc[3][2 + fx][r];
            }
    }
    if (p.y_absmax) publish_absmax(p.y_absmax, vmax, blockIdx.x * 4 + wave);
}

}  // namespace hfagp

using namespace hfagp;

namespace hfagp {

// Whether the streaming kernel takes this up-sampling layer, and its launch geometry.  `min_blocks`: the policy of the caller
// (upconv_fir.hip: the fused forms only where the launch fills the chip).
bool upfir_lean_plan(const HfagpModconvArgs* a, LeanParams& lp, long long min_blocks) {
    { const char* off = getenv("HFAGP_DEV_FIR_LEAN"); if (off && atoi(off) == 0) return false; }     // (developer / test switch, per call)
    if (a->mode != HFAGP_CONVT3X3_UP2 || a->Cin != LCIN || a->Cout % 64 != 0 || a->x_f16 || a->y_f16 || a->rgb_w || a->rgb_part) return false;
    const int kd = kind_of(a->precision);
    if (!(kd == 1 || kd == 2 || kd == 4 || kd == 5)) return false;
    lp = LeanParams{};
    lp.B = a->B; lp.H = a->H; lp.W = a->W; lp.Cout = a->Cout;
    lp.nstrip = (2 * a->W + 27) / 28;
    lp.nsteps = (2 * a->H + 2 + 7) / 8;
    const long long base = (long long)a->B * (a->Cout / 64) * lp.nstrip;
    // segments: about six rounds of two blocks per CU when the layer allows it, at least four steps per segment (measured, 32 -> 256
    // @128^2: B = 32 0.587 / 0.581 / 0.579 / 0.593 / 0.638 ms at 1 / 2 / 3 / 4 / 8 segments; B = 8 0.212 / 0.177 / 0.163 / 0.168 at 1 / 2 / 4 / 8)
    long long nseg = (6 * 2 * kNumCU + base - 1) / base;
    nseg = std::max(1ll, std::min<long long>(nseg, lp.nsteps / 4));
    { const char* dev = getenv("HFAGP_DEV_FIR_NSEG"); if (dev) nseg = std::max(1, std::min(atoi(dev), lp.nsteps)); }
    lp.nseg = (int)nseg;
    if (base * nseg < min_blocks) return false;
    return true;
}

int launch_upfir_lean(const HfagpModconvArgs* a, LeanParams& lp, hipStream_t s) {
    lp.x = a->x; lp.wt = a->wt; lp.styles = a->styles; lp.dcoef = a->dcoef; lp.noise = a->noise; lp.bias = a->bias;
    lp.x_absmax = a->x_absmax; lp.y_absmax = a->y_absmax; lp.y = a->y;
    lp.x_batch_stride = a->x_batch_stride;
    lp.act = a->act; lp.noise_strength = a->noise_strength; lp.alpha = a->alpha; lp.gain = a->gain; lp.clamp = a->clamp;
    HFAGP_REQUIRE(a->act != HFAGP_ACT_LRELU || (a->alpha >= 0.f && a->alpha <= 1.f), HFAGP_EUNSUPPORTED,
                  "upconv_fir (streaming kernel): leaky-ReLU slope %g outside [0, 1]", (double)a->alpha);
    HFAGP_REQUIRE(a->gain > 0.f, HFAGP_EUNSUPPORTED, "upconv_fir (streaming kernel): gain %g must be positive", (double)a->gain);
    const unsigned blocks = (unsigned)((long long)a->B * (a->Cout / 64) * lp.nstrip * lp.nseg);
    switch (kind_of(a->precision)) {
        case 1: upfir_lean_kernel<1><<<blocks, 256, 0, s>>>(lp); break;
        case 2: upfir_lean_kernel<2><<<blocks, 256, 0, s>>>(lp); break;
        case 5: upfir_lean_kernel<5><<<blocks, 256, 0, s>>>(lp); break;
        default: upfir_lean_kernel<4><<<blocks, 256, 0, s>>>(lp); break;
    }
    return check_launch("upconv_fir_fwd (streaming kernel)");
}

}  // namespace hfagp

// The up-sampling layer of a synthesis block in ONE pass over its output (hfagp_upconv_fir_fwd, include/hfagp.h):
//   y = bias_act(FIR(conv_transpose2d(x * s, W, stride 2)) * d + noise)          EG3D conv2d_resample(up = 2) + bias_act
// The two-kernel form (upconv_bf16_kernel -> y_t in HBM -> upfir_epilogue_kernel) moves the raw transposed-conv result
// y_t [B][2H+1][2W+1][Cout] to HBM in fp32 and back: x + 3 S bytes for a layer whose algorithmic traffic is x + S (S = the
// output tensor), and for the two super-resolution layers the FIR pass alone costs as much as the GEMM.  Here y_t never
// leaves the CU:
//   * a block (8 waves, 128 output channels) owns a COLUMN STRIP of 16 input columns (32 y_t columns) and walks down a
//     segment of 8-row tiles; per tile it runs the K loop of upconv_bf16_kernel (four output phases in four accumulator sets),
//     then hands the 16 x 32 x 128 y_t tile through LDS — 32 channels at a time, in the space of the A staging buffers —
//     to all 512 threads, which apply the separable 4 x 4 FIR, demodulation, noise, bias, leaky ReLU, gain and clamp and
//     store y;
//   * the last three y_t rows of a tile stay in LDS (48 KB) for the tile below, so inside a segment the vertical halo costs
//     nothing: a tile's round produces output rows [R0 - 2, R0 + 13];
//   * what a block cannot finish — the three output columns at each strip boundary, the three output rows at each segment
//     boundary — is finished by upfir_strip_kernel from thin raw strips (6 of 32 columns, 6 rows per segment) that the
//     blocks export: ~0.25 S of scratch traffic instead of 2 S.
// Same operand arithmetic, MFMA order and FIR / epilogue formulas as the two kernels it replaces.
#include <type_traits>
#include "conv16_common.h"

#ifndef HFAGP_FIR_UNROLL
#define HFAGP_FIR_UNROLL 1
#endif

namespace hfagp {

struct FirFuse {
    float* colstrip;     // [B][tiles_w - 1][Hp][6][Cout]: y_t columns 32 j - 3 .. 32 j + 2 around strip boundary j
    float* rowstrip;     // [B][nseg - 1][6][Wp][Cout]:    y_t rows R - 3 .. R + 2 around segment boundary R = 16 * first tile
    int nseg, Hp, Wp;    // Hp = 16 tiles_h, Wp = 32 tiles_w
};

// strip layouts (element offsets): a tile's export for one channel group is one contiguous run
//   colstrip [B][nb][side 2][C/32][Hp][3][32]: y_t column 32 j - 3 + cs (cs = 0..5, side = cs / 3) of strip boundary j
//   rowstrip [B][nb][C/32][6][Wp][32]:         y_t row R - 3 + rs (rs = 0..5) of segment boundary R
__host__ __device__ __forceinline__ size_t colstrip_at(int b, int nb, int j, int C, int Hp, int row, int cs, int ch) {
    return ((((((size_t)b * nb + (j - 1)) * 2 + cs / 3) * (C >> 5) + (ch >> 5)) * Hp + row) * 3 + cs % 3) * 32 + (ch & 31);
}
__host__ __device__ __forceinline__ size_t rowstrip_at(int b, int nb, int j, int C, int Wp, int rs, int col, int ch) {
    return (((((size_t)b * nb + (j - 1)) * (C >> 5) + (ch >> 5)) * 6 + rs) * Wp + col) * 32 + (ch & 31);
}

constexpr int FT_PLANE = 32 * 32;                         // floats per y_t row of the LDS tile: [col 32][channel 32]
constexpr int FT_BYTES = 16 * FT_PLANE * 4;               // 16 rows
constexpr int FW_BYTES = 4 * 3 * FT_PLANE * 4;            // windows: 4 channel groups x 3 rows

template <int KD, int IO>
__global__ void __launch_bounds__(512, 1) upconv_fir_kernel(const ConvParams p, const FirFuse ff) {
    constexpr int NP = kind_parts(KD);
    constexpr bool F16 = kind_f16(KD);
    constexpr bool XH = (IO & 1) != 0, YH = (IO & 2) != 0;       // fp16 storage of x / y (see modconv_bf16_kernel)
    constexpr int XB = XH ? 2 : 4;
    static_assert(IO == 0 || KD == 1, "fp16 storage goes with the single-pass fp16 arithmetic");
    constexpr int NW = 8, NTH = NW * 64;
    constexpr int TM = 2, TN = 1, WN = NW / 2, BM = 128, BNU = WN * TN * 32, PH = BM / PW, NITEM = 9, RB = NP == 1 ? 6 : 3;
    static_assert(BNU == 128 && PH == 8, "tile shape");
    constexpr int LPWB = RowPitch<NP>::value;
    constexpr int APOS = (PH + 2) * LPWB;
    constexpr int A_PART = APOS * APITCH, A_BUF = NP * A_PART;
    constexpr int REGION0 = 2 * A_BUF > FT_BYTES ? 2 * A_BUF : FT_BYTES;
    constexpr int I_GRP[NITEM] = {0, 0, 0, 0, 1, 1, 2, 2, 3};
    constexpr int I_PHASE[NITEM] = {0, 1, 2, 3, 0, 1, 0, 2, 0};
    constexpr int I_W[NITEM] = {0, 1, 3, 4, 6, 7, 2, 5, 8};
    constexpr int G_FIRST[4] = {0, 4, 6, 8};
    constexpr int G_OFF[4] = {(1 * LPWB + 1) * APITCH, (0 * LPWB + 1) * APITCH, (1 * LPWB + 0) * APITCH, 0};
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    char* As = lds_raw;                                           // K loop: [2][NP][APOS][48 B]
    float* T = reinterpret_cast<float*>(lds_raw);                 // epilogue: [16 rows][32 cols][32 ch] (aliases As)
    float* Wn = reinterpret_cast<float*>(lds_raw + REGION0);      // [4 groups][3 rows][32 cols][32 ch]: last rows of the tile above
    float* NZ = reinterpret_cast<float*>(lds_raw + REGION0 + FW_BYTES);         // [16 rows][32 cols] noise of the tile's outputs
    float* DB = NZ + 512;                                                        // [2][128]: demod coefficient, bias of the block's channels
    float* Ss = DB + 256;

    unsigned id = blockIdx.x;
    const int tiles_nu = p.Cout / BNU;
    const int tn_blk = __builtin_amdgcn_readfirstlane(id % tiles_nu);  id /= tiles_nu;
    const int tw = __builtin_amdgcn_readfirstlane(id % p.tiles_w);     id /= p.tiles_w;
    const int seg = __builtin_amdgcn_readfirstlane(id % ff.nseg);      id /= ff.nseg;
    const int b = __builtin_amdgcn_readfirstlane(id);
    const int n0 = tw * PW, co0 = tn_blk * BNU;
    const int t_begin = (p.tiles_h * seg) / ff.nseg, t_end = (p.tiles_h * (seg + 1)) / ff.nseg;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int h = lane >> 5, l31 = lane & 31;
    const int c_begin = 0, c_end = p.nchunks;

    const int npatch = p.ph * p.pw;
    constexpr int A_PER_T = ((PH + 2) * (PW + 2) * 4 + NTH - 1) / NTH;
    static_assert(A_PER_T == 2, "staging schedule: two slots per thread");
    const char* xb = reinterpret_cast<const char*>(p.x) + (long long)b * p.x_batch_stride * XB;
    for (int i = tid; i < p.Cin; i += NTH) Ss[i] = p.styles ? p.styles[(size_t)b * p.Cin + i] : 1.f;
    float sback = 1.f, sdown = 1.f;
    if constexpr (F16) sdown = style_range_guard(p.styles ? p.styles + (size_t)b * p.Cin : nullptr, p.Cin, lane, &sback, p.x_absmax);
    if (tid < 128) {
        DB[tid] = p.dcoef ? p.dcoef[(size_t)b * p.Cout + co0 + tid] : 1.f;
        DB[128 + tid] = p.bias ? p.bias[co0 + tid] : 0.f;
    }
    if (t_begin == 0)                    // top of the image: y_t row -1 is zero padding
        for (int i = tid; i < 4 * 3 * FT_PLANE; i += NTH) Wn[i] = 0.f;

    // ---- B fragments: one 32-column tile per wave, ring of RB items
    const char* wb = reinterpret_cast<const char*>(p.wt);
    const int cq8 = p.Cin >> 3;
    const int part_stride = 9 * cq8 * p.Cout;
    const unsigned bth = (unsigned)(h * p.Cout + co0 + wn * 32 + l31) * 16u;
    int apos[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int pidx = (wm * TM + tm) * 32 + l31;
        apos[tm] = ((pidx >> 4) * LPWB + (pidx & 15)) * APITCH + 16 * h;
    }
    constexpr int NPROD = NP == 1 ? 1 : NP == 2 ? 3 : 6;
    constexpr int PA[6] = {0, 1, 0, 1, 2, 0};
    constexpr int PB[6] = {0, 0, 1, 1, 0, 2};

    const int Ho2 = 2 * p.H, Wo2 = 2 * p.W;
    const int C0 = 32 * tw;
    float vmax = 0.f;

#pragma unroll 1
    for (int th = t_begin; th < t_end; ++th) {
        const int m0 = th * PH;
        // ---- A staging (as in upconv_bf16_kernel): patch rows m0-1 .. m0+PH-1, columns n0-1 .. n0+PW-1
        float4 ra[A_PER_T];
        unsigned aoff[A_PER_T];
        int lds_a[A_PER_T], soff[A_PER_T];
        float amask[A_PER_T];
#pragma unroll
        for (int k = 0; k < A_PER_T; ++k) {
            const int idx = min(tid + k * NTH, npatch * 4 - 1);
            const int pix = idx >> 2, q = idx & 3;
            lds_a[k] = ((pix / p.pw) * LPWB + pix % p.pw) * APITCH + 8 * q;
            const int iy = m0 - 1 + pix / p.pw, ix = n0 - 1 + pix % p.pw;
            const bool inside = iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
            aoff[k] = inside ? (unsigned)((iy * p.in_w + ix) * p.Cin + 4 * q) * (unsigned)XB : 0u;
            amask[k] = inside ? sdown : 0.f;
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
        auto store_a = [&](int chunk, auto buf_tag, auto k_tag) __attribute__((always_inline)) {
            constexpr int BUF = decltype(buf_tag)::value, k = decltype(k_tag)::value;
            const float m = amask[k];
            const float4 sv = *reinterpret_cast<const float4*>(Ss + chunk * CKB + soff[k]);
            uint2 parts[NP];
            if constexpr (XH) {
                const f32x2 s01 = {sv.x * m, sv.y * m}, s23 = {sv.z * m, sv.w * m};
                const f16x2 x01 = __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, ra[k].x));
                const f16x2 x23 = __builtin_bit_cast(f16x2, __builtin_bit_cast(unsigned, ra[k].y));
                parts[0] = make_uint2(__builtin_bit_cast(unsigned, x01 * __builtin_convertvector(s01, f16x2)),
                                      __builtin_bit_cast(unsigned, x23 * __builtin_convertvector(s23, f16x2)));
            } else {
                split4<KD>(make_float4(ra[k].x * (sv.x * m), ra[k].y * (sv.y * m), ra[k].z * (sv.z * m),
                                       ra[k].w * (sv.w * m)), parts);
            }
#pragma unroll
            for (int q = 0; q < NP; ++q)
                *reinterpret_cast<uint2*>(As + BUF * A_BUF + q * A_PART + lds_a[k]) = parts[q];
        };

        f32x16 acc[4][TM];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[f][tm][r] = 0.f;

        u32x4 bq[RB][NP];
        u32x4 af[2][TM][NP];
        auto issue_b = [&](int c, auto i_tag, auto slot_tag) __attribute__((always_inline)) {
            constexpr int I = decltype(i_tag)::value, SL = decltype(slot_tag)::value;
            const int cc = min(c, c_end - 1);
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const char* base = wb + (long long)(q * part_stride + (I_W[I] * cq8 + cc * 2) * p.Cout) * 16;
                bq[SL][q] = *reinterpret_cast<const u32x4*>(base + bth);
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
            constexpr int SL = (decltype(u_tag)::value * NITEM + I) % RB;
            if constexpr (I == G_FIRST[G] && G < 3) read_a(u_tag, std::integral_constant<int, G + 1>{});
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pr = 0; pr < NPROD; ++pr)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    acc[F][tm] = mfma16<F16>(af[G & 1][tm][PA[pr]], bq[SL][PB[pr]], acc[F][tm]);
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
        {
            __syncthreads();                                    // styles / zeroed windows in LDS; the FIR tile of the tile above is consumed
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
            load_a(min(c_begin + 1, c_end - 1));
            int cg = c_begin;
            for (; cg + 1 < c_end; cg += 2) {
                chunk(cg, std::integral_constant<int, 0>{});
                chunk(cg + 1, std::integral_constant<int, 1>{});
            }
            if (cg < c_end) chunk(cg, std::integral_constant<int, 0>{});
        }

        // ---- epilogue: the y_t tile (rows R0 .. R0+15, columns C0 .. C0+31) goes through LDS 32 channels at a time.
        // FIR thread = (channel quad q4, output column fcol, row half rh): 16-byte LDS reads and 16-byte global stores
        // (4 channels per lane: a quarter of the instructions of a channel-per-lane mapping, which is what bounds both).
        const int R0 = 16 * th;
        const bool have_window = th != t_begin || th == 0;
        const int oy_min = have_window ? max(R0 - 2, 0) : R0 + 1;
        const bool exp_top = th == t_begin && seg >= 1;                   // this tile's rows 0..2 -> segment boundary `seg`
        const bool exp_bot = th == t_end - 1 && seg + 1 < ff.nseg;        // rows 13..15 -> segment boundary `seg + 1`
        const int q4 = tid & 7, fcol = (tid >> 3) & 31, rh = tid >> 8;
        {   // noise of the tile's output pixels, pre-scaled: NZ[k][col] for output row R0 + k - 2, column C0 + col
            const int k = tid >> 5, col = tid & 31;
            const int oy = R0 + k - 2, ox = C0 + col;
            float nz = 0.f;
            if (p.noise && oy >= 0 && oy < Ho2 && ox < Wo2) nz = p.noise[(size_t)oy * Wo2 + ox] * p.noise_strength;
            NZ[tid] = nz;                   // (NZ is outside the staging buffers; read after the barriers below)
        }
        __syncthreads();                    // the K loop is done with the staging buffers: they become the FIR tile
#pragma unroll 1
        for (int g = 0; g < 4; ++g) {
            if (wn == g) {
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = 2 * (2 * (wm * TM + tm) + (r >> 3)) + (f >> 1);
                            const int col = 2 * (8 * ((r >> 2) & 1) + 4 * h + (r & 3)) + (f & 1);
                            const float v = F16 ? acc[f][tm][r] * sback : acc[f][tm][r];
                            T[row * FT_PLANE + col * 32 + l31] = v;
                        }
            }
            const int co4 = co0 + g * 32 + 4 * q4;
            __syncthreads();
            const float4 d = *reinterpret_cast<const float4*>(DB + g * 32 + 4 * q4);
            const float4 bs = *reinterpret_cast<const float4*>(DB + 128 + g * 32 + 4 * q4);
            {
                // source rows u = -3 .. 15 of this round: u < 0 -> window row u + 3 (the tile above), u >= 0 -> tile row u;
                // y_t columns fcol - 1 .. fcol + 2 of the strip (strip 0: column -1 IS the zero padding; columns >= 32
                // only feed outputs that belong to upfir_strip_kernel).
                // Arithmetic: the FIR taps [1 3 3 1] / 4 per axis are applied as (a + d) + 3 (b + c) with the 1 / 16 folded into
                // the demodulation coefficient, and gain * lrelu(z) as max(g z, g alpha z) with the gain folded into the
                // coefficient and the bias (g > 0, 0 <= alpha <= 1; linear: alpha = 1): 3 + 3 + 6 instructions per value instead
                // of 4 + 4 + 10 — the epilogue runs in the same waves as the GEMM, every instruction of it is exposed.
                const float* Wg = Wn + g * 3 * FT_PLANE;
                const float m0 = fcol >= 1 ? 1.f : 0.f;
                const int c0 = max(fcol - 1, 0) * 32 + 4 * q4, c1 = fcol * 32 + 4 * q4, c2 = min(fcol + 1, 31) * 32 + 4 * q4,
                          c3 = min(fcol + 2, 31) * 32 + 4 * q4;
                float4 raw;                                            // y_t[u][fcol] of the row hrow() read last
                auto hrow = [&](int u) __attribute__((always_inline)) -> float4 {
                    const float* rowp = u < 0 ? Wg + (u + 3) * FT_PLANE : T + u * FT_PLANE;
                    const float4 v0 = *reinterpret_cast<const float4*>(rowp + c0), v1 = *reinterpret_cast<const float4*>(rowp + c1);
                    const float4 v2 = *reinterpret_cast<const float4*>(rowp + c2), v3 = *reinterpret_cast<const float4*>(rowp + c3);
                    raw = v1;
                    float4 r;
                    r.x = fmaf(3.f, v1.x + v2.x, fmaf(v0.x, m0, v3.x));
                    r.y = fmaf(3.f, v1.y + v2.y, fmaf(v0.y, m0, v3.y));
                    r.z = fmaf(3.f, v1.z + v2.z, fmaf(v0.z, m0, v3.z));
                    r.w = fmaf(3.f, v1.w + v2.w, fmaf(v0.w, m0, v3.w));
                    return r;
                };
                const float gain = p.gain, alpha = p.act == HFAGP_ACT_LRELU ? p.alpha : 1.f;
                const float cl = p.clamp >= 0.f ? p.clamp : 3.0e38f;
                const float s16 = gain * 0.0625f;
                const float4 dg = make_float4(d.x * s16, d.y * s16, d.z * s16, d.w * s16);
                const float4 bg = make_float4(bs.x * gain, bs.y * gain, bs.z * gain, bs.w * gain);
                auto fin = [&](float vsum, float dgc, float nb) __attribute__((always_inline)) -> float {
                    const float z = fmaf(vsum, dgc, nb);
                    const float o = fmaxf(z, z * alpha);
                    return __builtin_amdgcn_fmed3f(o, -cl, cl);
                };
                const int ox = C0 + fcol;
                const bool col_ok = (fcol >= 1 || tw == 0) && fcol <= 29 && ox < Wo2;
                const int k0 = 8 * rh;                                 // this thread's output rows: k0 .. k0 + 7
                // raw strips for upfir_strip_kernel: the three columns on either side of a strip boundary leave from the
                // threads that have them in registers anyway (tile rows k0 .. k0 + 7 of column fcol)
                const int ebnd = fcol < 3 ? tw : tw + 1;               // boundary j sits between strips j - 1 and j
                const bool exp_col = (fcol < 3 || fcol >= 29) && ebnd >= 1 && ebnd <= p.tiles_w - 1;
                float* ecol = exp_col ? ff.colstrip + colstrip_at(b, p.tiles_w - 1, ebnd, p.Cout, ff.Hp, R0 + k0,
                                                                  fcol < 3 ? 3 + fcol : fcol - 29, co4) : nullptr;
                float4 h0 = hrow(k0 - 3), h1 = hrow(k0 - 2), h2 = hrow(k0 - 1);
#pragma unroll HFAGP_FIR_UNROLL
                for (int kk = 0; kk < 8; ++kk) {
                    const int k = k0 + kk;
                    const float4 h3 = hrow(k);
                    if (exp_col) *reinterpret_cast<float4*>(ecol + (size_t)kk * 3 * 32) = raw;
                    const int oy = R0 + k - 2;
                    if (col_ok && oy >= oy_min && oy < Ho2) {
                        const float nzg = NZ[k * 32 + fcol] * gain;
                        float4 o;
                        o.x = fin(fmaf(3.f, h1.x + h2.x, h0.x + h3.x), dg.x, nzg + bg.x);
                        o.y = fin(fmaf(3.f, h1.y + h2.y, h0.y + h3.y), dg.y, nzg + bg.y);
                        o.z = fin(fmaf(3.f, h1.z + h2.z, h0.z + h3.z), dg.z, nzg + bg.z);
                        o.w = fin(fmaf(3.f, h1.w + h2.w, h0.w + h3.w), dg.w, nzg + bg.w);
                        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
                        const size_t e = (((size_t)b * Ho2 + oy) * Wo2 + ox) * p.Cout + co4;
                        if constexpr (YH) {
                            typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                            const h4 hv = {(_Float16)o.x, (_Float16)o.y, (_Float16)o.z, (_Float16)o.w};
                            *reinterpret_cast<h4*>(reinterpret_cast<_Float16*>(p.out) + e) = hv;
                        } else {
                            *reinterpret_cast<float4*>(p.out + e) = o;
                        }
                    }
                    h0 = h1; h1 = h2; h2 = h3;
                }
            }
            {
                // ---- raw strips: the three rows on either side of a segment boundary (first / last tile of a segment only)
                if (exp_top || exp_bot) {
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int idx = tid + it * NTH;                // [3 rows][32 cols][8 quads]
                        if (idx < 3 * 32 * 8) {
                            const int q = idx & 7, col = (idx >> 3) & 31, j = idx >> 8;
                            const int ch = co0 + g * 32 + 4 * q;
                            if (exp_top)
                                *reinterpret_cast<float4*>(ff.rowstrip + rowstrip_at(b, ff.nseg - 1, seg, p.Cout, ff.Wp, 3 + j, C0 + col, ch)) =
                                    *reinterpret_cast<const float4*>(T + j * FT_PLANE + col * 32 + 4 * q);
                            if (exp_bot)
                                *reinterpret_cast<float4*>(ff.rowstrip + rowstrip_at(b, ff.nseg - 1, seg + 1, p.Cout, ff.Wp, j, C0 + col, ch)) =
                                    *reinterpret_cast<const float4*>(T + (13 + j) * FT_PLANE + col * 32 + 4 * q);
                        }
                    }
                }
            }
            __syncthreads();                // every thread is done with the tile and the window of group g
            if (wn == g && wm == 1) {       // rows 13, 14, 15 of this tile -> window of the tile below
                float* Wd = Wn + g * 3 * FT_PLANE;
#pragma unroll
                for (int fx = 0; fx < 2; ++fx)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int col = 2 * (8 * (q >> 2) + 4 * h + (q & 3)) + fx;
                        const float sc = F16 ? sback : 1.f;
                        Wd[0 * FT_PLANE + col * 32 + l31] = acc[2 + fx][1][q] * sc;          // row 13: m = 6, fy = 1
                        Wd[1 * FT_PLANE + col * 32 + l31] = acc[0 + fx][1][8 + q] * sc;      // row 14: m = 7, fy = 0
                        Wd[2 * FT_PLANE + col * 32 + l31] = acc[2 + fx][1][8 + q] * sc;      // row 15: m = 7, fy = 1
                    }
            }
        }
    }
    if (p.y_absmax) publish_absmax(p.y_absmax, vmax, blockIdx.x * 8 + wave);
}

// Finishes what the strip blocks could not: ROWMODE = false — the three output columns 32 j - 2 .. 32 j at every strip
// boundary j, all output rows, from colstrip; ROWMODE = true — the three output rows R - 2 .. R at every segment boundary,
// all output columns, from rowstrip.  One thread = 4 channels x one output column x a run of output rows (sliding window of
// horizontal sums, as upfir_epilogue_kernel).  Corner outputs are produced by both modes from the same values: identical.
struct StripFix {
    const float* src;
    const float* dcoef; const float* noise; const float* bias;
    void* y;
    float* y_absmax;
    int B, H, W, C;                 // H, W = INPUT resolution of the up-conv
    int nb;                         // boundaries per sample
    int Hp, Wp, nseg, tiles_h;
    int act; float noise_strength, alpha, gain, clamp;
};

constexpr int kFixRows = 8;

template <bool ROWMODE, bool H16>
__global__ void __launch_bounds__(256) upfir_strip_kernel(const StripFix a) {
    const int C4 = a.C >> 2;
    const int Ho = 2 * a.H, Wo = 2 * a.W;
    // work decomposition: (b, boundary, run, column, c4)
    const int ncol = ROWMODE ? Wo : 3;
    const int nrun = ROWMODE ? 1 : (Ho + kFixRows - 1) / kFixRows;
    const long long total = (long long)a.B * a.nb * nrun * ncol * C4;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= total) return;
    const int c4 = (int)(tid % C4);
    const int xi = (int)((tid / C4) % ncol);
    const int run = (int)((tid / ((long long)C4 * ncol)) % nrun);
    const int j = (int)((tid / ((long long)C4 * ncol * nrun)) % a.nb) + 1;         // boundary, 1-based
    const int b = (int)(tid / ((long long)C4 * ncol * nrun * a.nb));
    // source window: y_t(r, c) = src[(r - r_src0)][(c - c_src0)]
    int r_src0, c_src0, src_rows, src_cols, ox, oy_first, n_out;
    if constexpr (ROWMODE) {
        const int R = 16 * ((a.tiles_h * j) / a.nseg);
        r_src0 = R - 3; c_src0 = 0; src_rows = 6; src_cols = a.Wp;
        ox = xi; oy_first = R - 2; n_out = 3;
    } else {
        r_src0 = 0; c_src0 = 32 * j - 3; src_rows = a.Hp; src_cols = 6;
        ox = 32 * j - 2 + xi; oy_first = run * kFixRows; n_out = kFixRows;
    }
    if (ox >= Wo) return;
    const float f0 = 0.25f, f1 = 0.75f;
    int xs[4];                                                           // clamped source column of y_t column ox - 1 + q
    float wq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int cs = ox - 1 + q - c_src0;
        const bool ok = cs >= 0 && cs < src_cols && (ox - 1 + q) >= 0;
        xs[q] = min(max(cs, 0), src_cols - 1);
        wq[q] = ok ? ((q == 0 || q == 3) ? f0 : f1) : 0.f;
    }
    auto hrow = [&](int yin) -> float4 {                                 // yin = y_t row
        const int rs = yin - r_src0;
        const float m = (yin >= 0 && rs >= 0 && rs < src_rows) ? 1.f : 0.f;
        const int rc = min(max(rs, 0), src_rows - 1);
        float4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t e = ROWMODE ? rowstrip_at(b, a.nb, j, a.C, a.Wp, rc, xs[q], 4 * c4)
                                     : colstrip_at(b, a.nb, j, a.C, a.Hp, rc, xs[q], 4 * c4);
            v[q] = *reinterpret_cast<const float4*>(a.src + e);
        }
        float4 hsum;
        hsum.x = m * (wq[0] * v[0].x + wq[1] * v[1].x + wq[2] * v[2].x + wq[3] * v[3].x);
        hsum.y = m * (wq[0] * v[0].y + wq[1] * v[1].y + wq[2] * v[2].y + wq[3] * v[3].y);
        hsum.z = m * (wq[0] * v[0].z + wq[1] * v[1].z + wq[2] * v[2].z + wq[3] * v[3].z);
        hsum.w = m * (wq[0] * v[0].w + wq[1] * v[1].w + wq[2] * v[2].w + wq[3] * v[3].w);
        return hsum;
    };
    float4 d = make_float4(1.f, 1.f, 1.f, 1.f), bs = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.dcoef) d = reinterpret_cast<const float4*>(a.dcoef + (size_t)b * a.C)[c4];
    if (a.bias) bs = reinterpret_cast<const float4*>(a.bias)[c4];
    float vmax = 0.f;
    float4 h0 = hrow(oy_first - 1), h1 = hrow(oy_first), h2 = hrow(oy_first + 1);
#pragma unroll
    for (int k = 0; k < (ROWMODE ? 3 : kFixRows); ++k) {
        const int oy = oy_first + k;
        if (k >= n_out || oy >= Ho) break;
        const float4 h3 = hrow(oy + 2);
        if (oy >= 0) {
            float4 o;
            o.x = f0 * h0.x + f1 * h1.x + f1 * h2.x + f0 * h3.x;
            o.y = f0 * h0.y + f1 * h1.y + f1 * h2.y + f0 * h3.y;
            o.z = f0 * h0.z + f1 * h1.z + f1 * h2.z + f0 * h3.z;
            o.w = f0 * h0.w + f1 * h1.w + f1 * h2.w + f0 * h3.w;
            const float nz = a.noise ? a.noise[(size_t)oy * Wo + ox] * a.noise_strength : 0.f;
            o.x = lrelu_gain_clamp(o.x * d.x + nz + bs.x, a.act, a.alpha, a.gain, a.clamp);
            o.y = lrelu_gain_clamp(o.y * d.y + nz + bs.y, a.act, a.alpha, a.gain, a.clamp);
            o.z = lrelu_gain_clamp(o.z * d.z + nz + bs.z, a.act, a.alpha, a.gain, a.clamp);
            o.w = lrelu_gain_clamp(o.w * d.w + nz + bs.w, a.act, a.alpha, a.gain, a.clamp);
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            const size_t dst = (((size_t)b * Ho + oy) * Wo + ox) * C4 + c4;
            if constexpr (H16) {
                typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                const h4 hv = {(_Float16)o.x, (_Float16)o.y, (_Float16)o.z, (_Float16)o.w};
                reinterpret_cast<h4*>(a.y)[dst] = hv;
            } else {
                reinterpret_cast<float4*>(a.y)[dst] = o;
            }
        }
        h0 = h1; h1 = h2; h2 = h3;
    }
    if (a.y_absmax) publish_absmax(a.y_absmax, vmax, blockIdx.x * 4 + (threadIdx.x >> 6));
}

struct FusePlan {
    Plan pl;
    FirFuse ff;
    size_t col_bytes, row_bytes;
    bool ok;
};

// Segment count: enough blocks for ~8 rounds of the chip when the layer allows it, at least 2 tiles per segment (a segment
// of one tile has no window to reuse); unsupported (ok = false) when the launch could not fill the chip twice — small
// layers and small batches keep the two-kernel form with its split-K.
static int fuse_plan(const HfagpModconvArgs* a, FusePlan& fp) {
    fp.ok = false;
    fp.col_bytes = fp.row_bytes = 0;
    // (the scratch-size query has no y yet: validate() is not used here)
    HFAGP_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->Cin > 0 && a->Cout > 0, HFAGP_EBADARG, "upconv_fir: bad dims");
    HFAGP_REQUIRE(a->act == HFAGP_ACT_LINEAR || a->act == HFAGP_ACT_LRELU, HFAGP_EUNSUPPORTED, "upconv_fir: act %d", a->act);
    if (a->mode != HFAGP_CONVT3X3_UP2 || a->precision == HFAGP_PREC_F32 || a->precision == HFAGP_PREC_BF16X6) return HFAGP_OK;
    if (a->Cin % CKB != 0 || a->Cout % 128 != 0 || a->Cin > 512) return HFAGP_OK;
    HfagpModconvArgs t = *a;
    t.ksplit = 1;
    const int rc = make_plan(&t, fp.pl, 16);
    if (rc != HFAGP_OK) return rc;
    const ConvParams& p = fp.pl.p;
    const long long base = (long long)a->B * p.tiles_w * (a->Cout / 128);
    const char* dev_min = getenv("HFAGP_DEV_FIR_MIN_BLOCKS");      // (developer / test switches: read per call)
    const long long min_blocks = dev_min ? atoll(dev_min) : 2 * kNumCU;
    long long nseg = (8 * kNumCU + base - 1) / base;
    const long long max_seg = p.tiles_h >= 2 ? p.tiles_h / 2 : 1;
    if (nseg > max_seg) nseg = max_seg;
    if (nseg < 1) nseg = 1;
    { const char* dev = getenv("HFAGP_DEV_FIR_NSEG"); if (dev) nseg = std::max(1, std::min(atoi(dev), p.tiles_h)); }
    // measured (32 -> 256 @128^2): with fewer than one strip block per CU the segments get too short for the row window to pay
    // (B = 4: 0.26 ms against 0.22 for the two kernels, B = 8: a tie, B = 16: 0.83 vs 0.86, B = 32: 1.43 vs 1.60)
    if (base * nseg < min_blocks || (!dev_min && base < kNumCU)) return HFAGP_OK;
    fp.ff.nseg = (int)nseg;
    fp.ff.Hp = 16 * p.tiles_h;
    fp.ff.Wp = 32 * p.tiles_w;
    fp.col_bytes = (size_t)a->B * (p.tiles_w - 1) * fp.ff.Hp * 6 * a->Cout * sizeof(float);
    fp.row_bytes = (size_t)a->B * (nseg - 1) * 6 * fp.ff.Wp * a->Cout * sizeof(float);
    fp.ok = true;
    return HFAGP_OK;
}

template <int KD, int IO>
static int launch_fused(const FusePlan& fp, int cin, hipStream_t s) {
    constexpr int NP = kind_parts(KD);
    constexpr int A2 = 2 * NP * (8 + 2) * RowPitch<NP>::value * APITCH;
    const size_t lds = (size_t)(A2 > FT_BYTES ? A2 : FT_BYTES) + FW_BYTES + (size_t)(512 + 256 + cin + 8) * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&upconv_fir_kernel<KD, IO>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        set_error("upconv_fir: cannot raise dynamic LDS to %zu bytes: %s", lds, hipGetErrorString(e));
        return HFAGP_ELAUNCH;
    }
    const ConvParams& p = fp.pl.p;
    const unsigned blocks = (unsigned)((long long)p.B * fp.ff.nseg * p.tiles_w * (p.Cout / 128));
    upconv_fir_kernel<KD, IO><<<blocks, 512, lds, s>>>(p, fp.ff);
    return check_launch("upconv_fir_fwd");
}

}  // namespace hfagp

using namespace hfagp;

// the streaming kernel (upfir_lean.hip).  No fill-the-chip threshold: measured ahead of the two-kernel form at EVERY batch
// (32 -> 256 @128^2 -> 256^2, profiles/r06_upfir_lean_ab.log: B = 32 0.73 ms against 1.58 / 1.40 (strip kernel), B = 1 42 us against 66)
static bool lean_takes(const HfagpModconvArgs* a, LeanParams& lp) {
    if (!a || a->B <= 0 || a->H <= 0 || a->W <= 0 || a->Cout <= 0) return false;
    return upfir_lean_plan(a, lp, 1);
}

extern "C" size_t hfagp_upconv_fir_scratch_bytes(const HfagpModconvArgs* a) {
    LeanParams lp;
    if (lean_takes(a, lp)) return 256;             // (no scratch: any non-zero size says "supported")
    FusePlan fp;
    if (!a || fuse_plan(a, fp) != HFAGP_OK || !fp.ok) return 0;
    return fp.col_bytes + fp.row_bytes + 256;
}

extern "C" int hfagp_upconv_fir_fwd(const HfagpModconvArgs* a, void* scratch, void* stream) {
    HFAGP_REQUIRE(a && a->x && a->wt && a->y && scratch, HFAGP_EBADARG, "upconv_fir: null pointer");
    {
        LeanParams lp;
        if (lean_takes(a, lp)) return launch_upfir_lean(a, lp, (hipStream_t)stream);
    }
    FusePlan fp;
    int rc = fuse_plan(a, fp);
    if (rc != HFAGP_OK) return rc;
    HFAGP_REQUIRE(fp.ok, HFAGP_EUNSUPPORTED,
                  "upconv_fir: needs mode HFAGP_CONVT3X3_UP2, a two-part or single-pass 16-bit precision, Cin %% 16 == 0, Cin <= 512, "
                  "Cout %% 128 == 0 and a launch of at least 512 blocks (hfagp_upconv_fir_scratch_bytes() == 0 otherwise); got "
                  "mode %d precision %d B %d H %d W %d Cin %d Cout %d", a->mode, a->precision, a->B, a->H, a->W, a->Cin, a->Cout);
    HFAGP_REQUIRE(!(a->x_f16 || a->y_f16) || a->precision == HFAGP_PREC_F16, HFAGP_EUNSUPPORTED,
                  "upconv_fir: fp16 storage (x_f16 / y_f16) goes with precision HFAGP_PREC_F16");
    HFAGP_REQUIRE(!a->rgb_w && !a->rgb_part, HFAGP_EUNSUPPORTED, "upconv_fir: no fused toRGB on the up-sampling layer");
    hipStream_t s = (hipStream_t)stream;
    ConvParams& p = fp.pl.p;
    p.out = a->y;
    fp.ff.colstrip = reinterpret_cast<float*>(scratch);
    fp.ff.rowstrip = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + ((fp.col_bytes + 255) & ~(size_t)255));
    const int kd = kind_of(a->precision);
    const int io = (a->x_f16 ? 1 : 0) | (a->y_f16 ? 2 : 0);
    if (kd == 1) {
        switch (io) {
            case 0: rc = launch_fused<1, 0>(fp, a->Cin, s); break;
            case 1: rc = launch_fused<1, 1>(fp, a->Cin, s); break;
            case 2: rc = launch_fused<1, 2>(fp, a->Cin, s); break;
            default: rc = launch_fused<1, 3>(fp, a->Cin, s); break;
        }
    } else if (kd == 2) {
        rc = launch_fused<2, 0>(fp, a->Cin, s);
    } else {
        rc = launch_fused<4, 0>(fp, a->Cin, s);     // (F16X2 as well: the fused layer is only picked for short-K layers, same weight image)
    }
    if (rc != HFAGP_OK) return rc;
    StripFix f;
    f.dcoef = a->dcoef; f.noise = a->noise; f.bias = a->bias; f.y = a->y; f.y_absmax = a->y_absmax;
    f.B = a->B; f.H = a->H; f.W = a->W; f.C = a->Cout;
    f.Hp = fp.ff.Hp; f.Wp = fp.ff.Wp; f.nseg = fp.ff.nseg; f.tiles_h = p.tiles_h;
    f.act = a->act; f.noise_strength = a->noise_strength; f.alpha = a->alpha; f.gain = a->gain; f.clamp = a->clamp;
    const int C4 = a->Cout / 4;
    if (p.tiles_w > 1) {
        f.src = fp.ff.colstrip; f.nb = p.tiles_w - 1;
        const long long total = (long long)a->B * f.nb * ((2 * a->H + kFixRows - 1) / kFixRows) * 3 * C4;
        if (a->y_f16) upfir_strip_kernel<false, true><<<(unsigned)((total + 255) / 256), 256, 0, s>>>(f);
        else upfir_strip_kernel<false, false><<<(unsigned)((total + 255) / 256), 256, 0, s>>>(f);
        rc = check_launch("upconv_fir_fwd/column strips");
        if (rc != HFAGP_OK) return rc;
    }
    if (fp.ff.nseg > 1) {
        f.src = fp.ff.rowstrip; f.nb = fp.ff.nseg - 1;
        const long long total = (long long)a->B * f.nb * (2 * a->W) * C4;
        if (a->y_f16) upfir_strip_kernel<true, true><<<(unsigned)((total + 255) / 256), 256, 0, s>>>(f);
        else upfir_strip_kernel<true, false><<<(unsigned)((total + 255) / 256), 256, 0, s>>>(f);
        rc = check_launch("upconv_fir_fwd/row strips");
    }
    return rc;
}

// Weight gradient of the 3x3 modulated conv on v_mfma_f32_32x32x16_bf16 with split operands (gfx950):
//     dW[tap][ci][co] = sum_{b, pos} (x s)[b][pos + tap][ci] * g[b][pos][co]
// Same decomposition as wgrad.hip (M = ci, N = co, K = positions; a block owns a 64 x 64 (ci, co) tile for all nine
// taps and walks 4 x 16 position tiles; split-K slabs + wgrad_reduce_tiled_kernel), but the products are BF16X3: both
// operands are split into hi + lo bf16 parts (a gradient needs fp32's exponent range, which bf16 keeps) and a
// product is hi.hi + lo.hi + hi.lo accumulated in fp32 — 3 MFMAs of the 16x faster pipe instead of 8 fp32 MFMAs
// per 16 positions.
//
// A 16-bit MFMA operand is 8 K-CONTIGUOUS values per lane, and K = positions here, so the patches are TRANSPOSED on
// their way into LDS: x as [part][ci][patch position], g as [part][co][tile position] (a staging thread loads 4
// neighbouring positions x 4 channels and writes, per channel and part, one 8-byte run of 4 positions).  One MFMA
// K step = one row of 16 positions.  The tap shift dy picks the patch row; the shift dx moves the 8-position window
// by one element, which a 16-byte LDS read cannot do.  Round 5: the patch row is stored TWICE — copy A = patch columns
// 0..15, copy B = patch columns 2..17 — so that the windows dx = -1 (A) and dx = +1 (B) are both aligned 16-byte reads
// into even-aligned register quads (an MFMA operand must start on an even register: with one copy + a fifth dword the
// dx = +1 window cost four moves per part and group on top of the read) and dx = 0 is four v_alignbit_b32 / v_perm_b32 of
// the two; no read has a bank conflict (the 4-byte fifth-dword reads were 4-way conflicted: the row pitch has to be a
// multiple of 4 dwords, so 32 rows hit 8 banks — 60 % of the round-4 kernel's LDS cycles).
// LDS rows are PERMUTED: channel 4q + e of the tile lives in row 16 e + q, so that the 16 staging lanes of a column
// group (q = 0..15, one float4 = channels 4q..4q+3 each) write 16 consecutive rows (row pitch 100 / 36 dwords = 36 mod 64:
// conflict-free 8-byte writes) and the 32 lanes of an MFMA operand read 32 consecutive rows (conflict-free);
// MFMA row / column l of wave w therefore stands for channel 4 (l & 15) + 2 w + (l >> 4) of the tile.
#include <type_traits>
#include "common.h"

namespace hfagp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// The kernel is written in ROLES: the SHIFTED operand "a" (staged with a one-position halo, its channels are the MFMA
// rows) and the PLAIN operand "b" (its channels are the MFMA columns).  3x3 conv: a = x*s (shift = tap offset), b = g.
// Parity launches of the up-conv (SWAP): a = the parity image of the y_t gradient (shift 0 / +1), b = x*s.
struct Wg16Params {
    const float* a; const float* b; const float* styles;       // styles [B][channels of x] or null
    float* slabs;                       // [ksplit][Cout][Cin][9]: the parameter layout (tap = 3 (dy) + dx)
    int B, aH, aW, aC, bH, bW, bC;      // image extents / channel counts of the two operands (bH x bW = position grid)
    int Cin, Cout, tiles_h, tiles_w, ksplit;
};

constexpr int QH = 4, QW = 16;                   // position tile (4 rows: 108 MFMAs per wave per tile)
constexpr int XR = QH + 2;                       // x patch: 6 rows of 18 columns (halo 1 + 16 + 1)
constexpr int CT = 64;                           // ci / co tile
constexpr int XROW = 32;                         // bytes per patch row of one copy (16 columns)
constexpr int XCOPY = XR * XROW;                 // 192: copy A at 0, copy B at XCOPY
constexpr int XPITCH = 2 * XCOPY + 16;           // bytes per ci row of one part (400: 100 dwords = 36 mod 64, conflict-free b128);
                                                 // the 16 pad bytes absorb the writes of unit columns a copy does not hold
constexpr int GPITCH = QH * QW * 2 + 16;         // bytes per co row of one part (144: 36 dwords)
constexpr int XPART = CT * XPITCH, GPART = CT * GPITCH;
constexpr int BUF = 2 * XPART + 2 * GPART;       // one stage: x hi, x lo, g hi, g lo  (69 632 B; two stages)

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// four floats (consecutive positions of one channel) -> hi and lo part, two dwords each
__device__ __forceinline__ void split_run(float a, float b, float c, float d, uint2& hi, uint2& lo) {
    hi = make_uint2(pk_bf16(a, b), pk_bf16(c, d));
    lo = make_uint2(pk_bf16(a - __builtin_bit_cast(float, hi.x << 16), b - __builtin_bit_cast(float, hi.x & 0xffff0000u)),
                    pk_bf16(c - __builtin_bit_cast(float, hi.y << 16), d - __builtin_bit_cast(float, hi.y & 0xffff0000u)));
}

constexpr int popc3(int m) { return (m & 1) + ((m >> 1) & 1) + ((m >> 2) & 1); }
constexpr int rank3(int m, int i) { return popc3(m & ((1 << i) - 1)); }      // index of set bit i among the set bits
constexpr int nth3(int m, int j) {                                          // position of the j-th set bit
    int seen = 0;
    for (int i = 0; i < 3; ++i) if ((m >> i) & 1) { if (seen == j) return i; ++seen; }
    return 0;
}

// Epilogue of both kernels (round 5): the block's nine 64 x 64 accumulator tiles leave in the PARAMETER layout
// slab[co][ci][tap] — through LDS (the staging buffers are free by then): lane (co, rows ci) writes [co][ci * 9 + tap] with a
// row pitch of 64 * 9 + 1 floats (consecutive co -> consecutive banks), the block reads each co row back as 576 consecutive
// floats and stores them as one contiguous run.  The split-K reducer is then a plain sum of identical layouts (the slabs used to
// be [tap][ci][co] and the reducer a transposing gather: 28 us for 38-47 MB, launch after launch).
constexpr int EPI_PITCH = CT * 9 + 1;
constexpr int EPI_BYTES = CT * EPI_PITCH * 4;            // 147 712
__device__ __forceinline__ void store_slab_final(const f32x16 (&acc)[9], char* lds, float* slab, int Cin, int ci0, int co0,
                                                 int wi, int wj, int h, int l31, int tid) {
    const int run = min(CT, Cin - ci0) * 9;                  // floats per co row (a partial ci tile: Cin = 32 of the first SR layer)
    float* t32 = reinterpret_cast<float*>(lds);
    __syncthreads();                                         // every wave is done with the staging buffers
    const int cb = 4 * (l31 & 15) + 2 * wj + (l31 >> 4);     // MFMA column -> co of the tile (row permutation of the staging)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int ca = 4 * (rr & 15) + 2 * wi + (rr >> 4);               // MFMA row -> ci of the tile
            t32[cb * EPI_PITCH + ca * 9 + t] = acc[t][r];
        }
    __syncthreads();
    for (int c = 0; c < CT; ++c) {
        float* dst = slab + ((size_t)(co0 + c) * Cin + ci0) * 9;
        const float* src = t32 + c * EPI_PITCH;
        if (tid < run) dst[tid] = src[tid];
        if (tid + 256 < run) dst[tid + 256] = src[tid + 256];
        if (tid + 512 < run) dst[tid + 512] = src[tid + 512];
    }
}

// DYM / DXM: bit i set = patch-row / window offset i is used (shift i - 1).  3x3: 7, 7 (nine taps).  Parity images of the
// up-conv: shifts {0, +1} = bits 1, 2 (6) or {0} (2).  SWAP: operand a is the gradient, b is x (styles go to b, and
// the tile is stored transposed into the [Cin][Cout] slab).
//
// Schedule (round 5).  One wave per SIMD (144 accumulator registers), so nothing but this wave's own instruction stream can
// fill the matrix pipe's shadow.  PMC of the round-4 kernel (profiles/r05_pmc/wgrad_256_256_at256.txt): 5.4 vector-ALU
// instructions per MFMA, MFMA pipe 0.39 busy, 60 % of the LDS cycles bank conflicts.  That kernel ran the K loop of tile u
// (108 MFMAs), THEN converted + wrote tile u+1 (≈ 270 VALU + 24 LDS writes with the pipe idle), then a barrier.  Now
//   * the staging work of tile u+1 is cut into 12 pieces (unit x channel: one split_run + its LDS writes) and piece g is
//     issued inside group g of the K loop (a group = one patch row of one K step: 4 LDS reads, 8 v_perm, 9 MFMAs), the LDS
//     reads of group g+1 in front of the MFMAs of group g (a sched_barrier per group pins that);
//   * which needs tile u+1's loads to have LANDED when the K loop of tile u starts: the loads run a whole tile ahead of the
//     conversion (two register sets, the tile loop unrolled by two so that both are statically indexed);
//   * the loop body has NO branch: hipcc's s_waitcnt insertion counts outstanding loads exactly only along straight-line
//     code (with `if (u + 2 < u_end) fetch(...)` every conversion piece waited for vmcnt(0), i.e. for the loads just
//     issued).  A block therefore always runs an even number of phases; tiles past its range are NULL tiles — every load
//     out of range — and contribute zeros, and every phase converts "the next tile" whether or not there is one;
//   * raw buffer loads: a position is a 32-bit offset = scalar tile origin + per-thread constant, and an out-of-image (or
//     null-tile) position gets an offset past the end of the resource, for which the hardware returns zeros: no masks (12
//     registers and 48 multiplies per tile), no 64-bit address arithmetic, no memory traffic for what is not there.
template <int DYM, int DXM, bool SWAP>
__global__ void __launch_bounds__(256, 1) wgrad_bf16_kernel(const Wg16Params p) {
    constexpr int NDX = popc3(DXM), NDY = popc3(DYM), NT = NDY * NDX;
    constexpr int NG = QH * NDY;                      // groups of the K loop
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;     // wave tile: a-channel rows 32*wi.., b-channel cols 32*wj..
    const int h = lane >> 5, l31 = lane & 31;
    const int ci0 = blockIdx.x * CT, co0 = blockIdx.y * CT, ks = blockIdx.z;      // a-channel / b-channel tile origins

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int units = p.B * p.tiles_h * p.tiles_w;
    const int u_begin = (int)(((long long)units * ks) / p.ksplit), u_end = (int)(((long long)units * (ks + 1)) / p.ksplit);

    // ---- staging.  x: 6 rows x 5 column groups (4 columns each, 20 >= 18) x 16 channel groups = 480 units, thread t
    // owns units t and 256 + t (t < 224; the others repeat the last unit: same data, same addresses); g: 4 rows x 4 column
    // groups x 16 channel groups = 256 units, one per thread.  Unit = 4 float4 loads (one per column, 4 channels each);
    // nothing in fetch() consumes a loaded value, so the loads stay in flight under the MFMAs.
    float4 rx[2][2][4], rg[2][4], sx[2][2], sg[2];       // [register set][...]
    const int ux0 = tid, ux1 = min(tid + 256, XR * 5 * 16 - 1);
    const int xq0 = ux0 & 15, xcg0 = (ux0 >> 4) % 5, xrow0 = (ux0 >> 4) / 5;
    const int xq1 = ux1 & 15, xcg1 = (ux1 >> 4) % 5, xrow1 = (ux1 >> 4) / 5;
    const int gq = tid & 15, gcg = (tid >> 4) & 3, grow = tid >> 6;
    // LDS byte offsets of a unit's writes inside a stage, channel e = 0, hi part (+ 16 e XPITCH, + XPART: immediates).
    // x unit (row, cg) holds patch columns 4cg .. 4cg+3.  Copy A keeps columns 0..15: an 8-byte write for cg <= 3; copy B
    // keeps columns 2..17 at index col - 2: the pair's first dword goes to dword 2cg - 1 (cg >= 1), its second to dword
    // 2cg (cg <= 3).  What a copy does not hold is written into the row's 16 pad bytes instead (no branch in a piece).
    auto x_offsets = [&](int row, int cg, int q, int& oa, int& ob0, int& ob1) {
        const int base = q * XPITCH;
        oa = base + (cg <= 3 ? row * XROW + 8 * cg : 2 * XCOPY);
        ob0 = base + (cg >= 1 ? XCOPY + row * XROW + 4 * (2 * cg - 1) : 2 * XCOPY + 8);
        ob1 = base + (cg <= 3 ? XCOPY + row * XROW + 8 * cg : 2 * XCOPY + 12);
    };
    int oa0, ob00, ob10, oa1, ob01, ob11;
    x_offsets(xrow0, xcg0, xq0, oa0, ob00, ob10);
    x_offsets(xrow1, xcg1, xq1, oa1, ob01, ob11);
    const int og = 2 * XPART + gq * GPITCH + (grow * QW + 4 * gcg) * 2;

    // Loads are raw buffer loads: the image of sample b is a buffer resource (4 scalar registers), a position is a 32-bit byte
    // offset = (scalar tile origin) + (per-thread constant).
    const int xoff0 = ((xrow0 * p.aW + 4 * xcg0) * p.aC + 4 * xq0) * 4, xoff1 = ((xrow1 * p.aW + 4 * xcg1) * p.aC + 4 * xq1) * 4;
    const int goff = ((grow * p.bW + 4 * gcg) * p.bC + 4 * gq) * 4;
    const unsigned abytes = (unsigned)(p.aH * p.aW * p.aC) * 4u, bbytes = (unsigned)(p.bH * p.bW * p.bC) * 4u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.styles ? p.styles : p.a), 0, p.styles ? (unsigned)(p.B * (SWAP ? p.bC : p.aC)) * 4u : 0u, 0x00020000);
    // (no styles: the resource is empty, the load returns zeros and `one` = 1 is added — a select on a uniform condition
    // became a branch around the load, which costs hipcc its exact count of the loads in flight)
    const float one = p.styles ? 0.f : 1.f;
    constexpr unsigned OOB = 0xfffffff0u;
    auto fetch = [&](int u, auto set_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value;
        const bool live = u < u_end;                         // (a null tile: every offset out of range)
        const int uc = live ? u : u_end - 1;
        const int tw = uc % p.tiles_w, th = (uc / p.tiles_w) % p.tiles_h, b = uc / (p.tiles_w * p.tiles_h);
        const int m0 = th * QH, n0 = tw * QW;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.a + (size_t)b * p.aH * p.aW * p.aC + ci0), 0, live ? abytes - 4u * ci0 : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.b + (size_t)b * p.bH * p.bW * p.bC + co0), 0, live ? bbytes - 4u * co0 : 0u, 0x00020000);
        const int abase_t = (((m0 - 1) * p.aW + n0 - 1) * p.aC) * 4, bbase_t = ((m0 * p.bW + n0) * p.bC) * 4;      // scalar
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int row = k ? xrow1 : xrow0, cg = k ? xcg1 : xcg0, q = k ? xq1 : xq0;
            const bool rowok = (unsigned)(m0 - 1 + row) < (unsigned)p.aH;
            if constexpr (!SWAP) {
                const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((b * p.aC + ci0 + 4 * q) * 4), 0, 0));
                sx[S][k] = make_float4(v.x + one, v.y + one, v.z + one, v.w + one);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool ok = rowok && (unsigned)(n0 - 1 + 4 * cg + c) < (unsigned)p.aW;
                const unsigned off = (unsigned)(abase_t + c * p.aC * 4 + (k ? xoff1 : xoff0));
                rx[S][k][c] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra, ok ? off : OOB, 0, 0));
            }
        }
        {
            const bool rowok = m0 + grow < p.bH;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool ok = rowok && n0 + 4 * gcg + c < p.bW;
                const unsigned off = (unsigned)(bbase_t + c * p.bC * 4 + goff);
                rg[S][c] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rb, ok ? off : OOB, 0, 0));
            }
            if constexpr (SWAP) {
                const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((b * p.bC + co0 + 4 * gq) * 4), 0, 0));
                sg[S] = make_float4(v.x + one, v.y + one, v.z + one, v.w + one);
            }
        }
    };
    // piece P of the conversion of register set S into LDS stage `buf`: P = 0..3 channel e of x unit 0, 4..7 of x unit 1,
    // 8..11 of the g unit.  One split_run + its LDS writes (x: 8-byte to copy A, two 4-byte to copy B, per part).
    auto commit_piece = [&](int buf, auto set_tag, auto piece_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value, P = decltype(piece_tag)::value, e = P & 3;
        char* base = lds + buf * BUF;
        uint2 hi, lo;
        if constexpr (P < 8) {
            constexpr int k = P >> 2;
            const float* f0 = &rx[S][k][0].x; const float* f1 = &rx[S][k][1].x;
            const float* f2 = &rx[S][k][2].x; const float* f3 = &rx[S][k][3].x;
            if constexpr (!SWAP) {
                const float sv = e == 0 ? sx[S][k].x : e == 1 ? sx[S][k].y : e == 2 ? sx[S][k].z : sx[S][k].w;
                split_run(f0[e] * sv, f1[e] * sv, f2[e] * sv, f3[e] * sv, hi, lo);
            } else {
                split_run(f0[e], f1[e], f2[e], f3[e], hi, lo);
            }
            char* row = base + 16 * e * XPITCH;
            *reinterpret_cast<uint2*>(row + (k ? oa1 : oa0)) = hi;
            *reinterpret_cast<uint2*>(row + (k ? oa1 : oa0) + XPART) = lo;
            *reinterpret_cast<unsigned*>(row + (k ? ob01 : ob00)) = hi.x;
            *reinterpret_cast<unsigned*>(row + (k ? ob11 : ob10)) = hi.y;
            *reinterpret_cast<unsigned*>(row + (k ? ob01 : ob00) + XPART) = lo.x;
            *reinterpret_cast<unsigned*>(row + (k ? ob11 : ob10) + XPART) = lo.y;
        } else {
            const float* f0 = &rg[S][0].x; const float* f1 = &rg[S][1].x; const float* f2 = &rg[S][2].x; const float* f3 = &rg[S][3].x;
            if constexpr (SWAP) {
                const float sv = e == 0 ? sg[S].x : e == 1 ? sg[S].y : e == 2 ? sg[S].z : sg[S].w;
                split_run(f0[e] * sv, f1[e] * sv, f2[e] * sv, f3[e] * sv, hi, lo);
            } else {
                split_run(f0[e], f1[e], f2[e], f3[e], hi, lo);
            }
            char* dst = base + og + 16 * e * GPITCH;
            *reinterpret_cast<uint2*>(dst) = hi;
            *reinterpret_cast<uint2*>(dst + GPART) = lo;
        }
    };
    auto commit_all = [&](int buf, auto set_tag) __attribute__((always_inline)) {
        commit_piece(buf, set_tag, std::integral_constant<int, 0>{});  commit_piece(buf, set_tag, std::integral_constant<int, 1>{});
        commit_piece(buf, set_tag, std::integral_constant<int, 2>{});  commit_piece(buf, set_tag, std::integral_constant<int, 3>{});
        commit_piece(buf, set_tag, std::integral_constant<int, 4>{});  commit_piece(buf, set_tag, std::integral_constant<int, 5>{});
        commit_piece(buf, set_tag, std::integral_constant<int, 6>{});  commit_piece(buf, set_tag, std::integral_constant<int, 7>{});
        commit_piece(buf, set_tag, std::integral_constant<int, 8>{});  commit_piece(buf, set_tag, std::integral_constant<int, 9>{});
        commit_piece(buf, set_tag, std::integral_constant<int, 10>{}); commit_piece(buf, set_tag, std::integral_constant<int, 11>{});
    };

    // per-lane fragment bases (bytes inside a stage): A = LDS row 32 wi + l31, columns 8h..; B = LDS row 32 wj + l31
    const int abase = (32 * wi + l31) * XPITCH + 8 * h * 2;
    const int bbase = 2 * XPART + (32 * wj + l31) * GPITCH + 8 * h * 2;

    // fragments of one group as they come out of LDS: copy A / copy B window of the patch row, hi and lo part
    struct Raw { u32x4 ha, hb, la, lb; };
    auto read_a = [&](const char* st, int prow) __attribute__((always_inline)) {
        const char* ar = st + abase + prow * XROW;
        Raw r;
        r.ha = *reinterpret_cast<const u32x4*>(ar);
        r.hb = *reinterpret_cast<const u32x4*>(ar + XCOPY);
        r.la = *reinterpret_cast<const u32x4*>(ar + XPART);
        r.lb = *reinterpret_cast<const u32x4*>(ar + XPART + XCOPY);
        return r;
    };
    // the MFMAs of one group: K step kr (tile row), patch row kr + dy
    auto mfmas = [&](const Raw& w, auto dy_tag, const u32x4& bh, const u32x4& bl) __attribute__((always_inline)) {
        constexpr int dy = decltype(dy_tag)::value;
        u32x4 ah[3], al[3];                       // the three dx windows (start column 8h + dx)
        ah[0] = w.ha; al[0] = w.la; ah[2] = w.hb; al[2] = w.lb;
        if constexpr ((DXM >> 1) & 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {         // columns 8h+1+2e, 8h+2+2e = high half of A's dword e, low half of B's
                ah[1][e] = __builtin_amdgcn_alignbit(w.hb[e], w.ha[e], 16);
                al[1][e] = __builtin_amdgcn_alignbit(w.lb[e], w.la[e], 16);
            }
        }
        constexpr int trow = rank3(DYM, dy) * NDX;    // first tap slot of this row
        // product-major: the three MFMAs of one accumulator are up to two other MFMAs apart
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
            if ((DXM >> dx) & 1)
                acc[trow + rank3(DXM, dx)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[dx]), __builtin_bit_cast(bf16x8, bh), acc[trow + rank3(DXM, dx)], 0, 0, 0);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
            if ((DXM >> dx) & 1)
                acc[trow + rank3(DXM, dx)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[dx]), __builtin_bit_cast(bf16x8, bh), acc[trow + rank3(DXM, dx)], 0, 0, 0);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
            if ((DXM >> dx) & 1)
                acc[trow + rank3(DXM, dx)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[dx]), __builtin_bit_cast(bf16x8, bl), acc[trow + rank3(DXM, dx)], 0, 0, 0);
    };
    // the pieces of the next tile's conversion that ride in group G (12 pieces over NG groups: piece P goes to group P % NG)
    auto pieces_of = [&](int buf, auto set_tag, auto g_tag) __attribute__((always_inline)) {
        constexpr int G = decltype(g_tag)::value;
        if constexpr (G < 12) commit_piece(buf, set_tag, std::integral_constant<int, (G < 12 ? G : 0)>{});
        if constexpr (G + NG < 12) commit_piece(buf, set_tag, std::integral_constant<int, (G + NG < 12 ? G + NG : 0)>{});
        if constexpr (G + 2 * NG < 12) commit_piece(buf, set_tag, std::integral_constant<int, (G + 2 * NG < 12 ? G + 2 * NG : 0)>{});
    };
    // K loop of the tile staged in `cur`, groups in (kr, dy) order, while register set S (the next tile) is converted into
    // the other stage
    auto tile = [&](int cur, auto set_tag) __attribute__((always_inline)) {
        const char* st = lds + cur * BUF;
        Raw cur_a = read_a(st, nth3(DYM, 0));
        u32x4 bh = *reinterpret_cast<const u32x4*>(st + bbase);
        u32x4 bl = *reinterpret_cast<const u32x4*>(st + bbase + GPART);
        auto step = [&](auto g_tag) __attribute__((always_inline)) {
            constexpr int G = decltype(g_tag)::value;
            constexpr int kr = G / NDY, dy = nth3(DYM, G % NDY);
            constexpr int G1 = G + 1, kr1 = G1 / NDY, dy1 = nth3(DYM, G1 % NDY);
            Raw nxt = cur_a;
            u32x4 nbh = bh, nbl = bl;
            if constexpr (G1 < NG) {
                nxt = read_a(st, kr1 + dy1);
                if constexpr (kr1 != kr) {
                    nbh = *reinterpret_cast<const u32x4*>(st + bbase + kr1 * QW * 2);
                    nbl = *reinterpret_cast<const u32x4*>(st + bbase + GPART + kr1 * QW * 2);
                }
            }
            mfmas(cur_a, std::integral_constant<int, dy>{}, bh, bl);
            pieces_of(cur ^ 1, set_tag, g_tag);
            __builtin_amdgcn_sched_barrier(0);
            cur_a = nxt; bh = nbh; bl = nbl;
        };
        step(std::integral_constant<int, 0>{});
        if constexpr (NG > 1) step(std::integral_constant<int, (NG > 1 ? 1 : 0)>{});
        if constexpr (NG > 2) step(std::integral_constant<int, (NG > 2 ? 2 : 0)>{});
        if constexpr (NG > 3) step(std::integral_constant<int, (NG > 3 ? 3 : 0)>{});
        if constexpr (NG > 4) step(std::integral_constant<int, (NG > 4 ? 4 : 0)>{});
        if constexpr (NG > 5) step(std::integral_constant<int, (NG > 5 ? 5 : 0)>{});
        if constexpr (NG > 6) step(std::integral_constant<int, (NG > 6 ? 6 : 0)>{});
        if constexpr (NG > 7) step(std::integral_constant<int, (NG > 7 ? 7 : 0)>{});
        if constexpr (NG > 8) step(std::integral_constant<int, (NG > 8 ? 8 : 0)>{});
        if constexpr (NG > 9) step(std::integral_constant<int, (NG > 9 ? 9 : 0)>{});
        if constexpr (NG > 10) step(std::integral_constant<int, (NG > 10 ? 10 : 0)>{});
        if constexpr (NG > 11) step(std::integral_constant<int, (NG > 11 ? 11 : 0)>{});
    };
    // phase: tile u sits in stage S (converted from register set S, which is free again); register set 1 - S holds tile u + 1
    auto phase = [&](int u, auto s_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(s_tag)::value;
        fetch(u + 2, s_tag);
        tile(S, std::integral_constant<int, 1 - S>{});
        __syncthreads();
    };

    fetch(u_begin, std::integral_constant<int, 0>{});
    fetch(u_begin + 1, std::integral_constant<int, 1>{});
    commit_all(0, std::integral_constant<int, 0>{});
    __syncthreads();
    for (int u = u_begin; u < u_end; u += 2) {               // (an odd tile count ends with a null tile: zeros)
        phase(u, std::integral_constant<int, 0>{});
        phase(u + 1, std::integral_constant<int, 1>{});
    }
    // ---- slab [Cout][Cin][9] (the parameter layout): C/D layout row = (r&3) + 8*(r>>2) + 4*h (operand-a channel = ci),
    // col = lane&31 (operand-b channel = co)
    static_assert(NT == 9 && !SWAP, "the final-layout epilogue is written for the nine-tap, un-swapped kernel");
    store_slab_final(acc, lds, p.slabs + (size_t)ks * 9 * p.Cin * p.Cout, p.Cin, ci0, co0, wi, wj, h, l31, tid);
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of the UP-SAMPLING conv in one kernel (round 5; before: four launches of the kernel above, one per parity image
// of the y_t gradient, each staging the x tile again — 1.3 ms of the generator-tuned step at 0.11-0.16 of the MFMA ceiling).
//   dW[ky][kx][ci][co] = sum_{b,i,j} (x s)[b][i][j][ci] * g_t[b][2i + ky][2j + kx][co]
//                      = sum_{b,m,n} (x s)[b][m - sy][n - sx][ci] * g_par[p][q][b][m][n][co],   ky = p + 2 sy, kx = q + 2 sx
// on the (H+1) x (W+1) grid of a parity image: the SHIFTED operand is x (shifts -1 / 0: halo on top and on the left, staged
// ONCE per position tile), the plain operand is the tile of ONE parity image, and a tile is four PHASES — parity (1,1): 1 tap,
// (0,1) and (1,0): 2 taps, (0,0): 4 taps — that share the nine accumulators of the 3x3 kernel (tap slot = 3 ky + kx).
// Same pipeline as above, per phase instead of per tile: the gradient tile of (tile u+1, phase f) is loaded at the start of (tile u,
// phase f) (four register sets: a whole tile of latency) and converted inside the K loop of the phase before its own (4 pieces); the
// x patch of tile u+1 is loaded at the start of tile u and converted inside its last, 4-tap phase (8 pieces).  The loop body (one tile = four
// phases) has no branch; phases past the block's range load zeros.
struct WgUpParams {
    const float* x; const float* g; const float* styles;      // g = [2][2][B][H+1][W+1][Cout]
    float* slabs;                                             // [ksplit][Cout][Cin][9]
    int B, H, W, Cin, Cout, tiles_h, tiles_w, ksplit;
};
constexpr int UBX = 2 * XPART, UBG = 2 * GPART;               // one x stage (hi, lo), one g stage

__global__ void __launch_bounds__(256, 1) wgrad_up_bf16_kernel(const WgUpParams p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];     // [x stage 0][x stage 1][g stage 0][g stage 1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;
    const int ci0 = blockIdx.x * CT, co0 = blockIdx.y * CT, ks = blockIdx.z;
    const int gH = p.H + 1, gW = p.W + 1;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int units = p.B * p.tiles_h * p.tiles_w;
    const int u_begin = (int)(((long long)units * ks) / p.ksplit), u_end = (int)(((long long)units * (ks + 1)) / p.ksplit);

    float4 rx[2][4], sx[2], rg[4][4];                          // x: one register set (two units); g: [set = phase of the tile][column]
    const int ux0 = tid, ux1 = min(tid + 256, XR * 5 * 16 - 1);
    const int xq0 = ux0 & 15, xcg0 = (ux0 >> 4) % 5, xrow0 = (ux0 >> 4) / 5;
    const int xq1 = ux1 & 15, xcg1 = (ux1 >> 4) % 5, xrow1 = (ux1 >> 4) / 5;
    const int gq = tid & 15, gcg = (tid >> 4) & 3, grow = tid >> 6;
    auto x_offsets = [&](int row, int cg, int q, int& oa, int& ob0, int& ob1) {      // (as in the 3x3 kernel)
        const int base = q * XPITCH;
        oa = base + (cg <= 3 ? row * XROW + 8 * cg : 2 * XCOPY);
        ob0 = base + (cg >= 1 ? XCOPY + row * XROW + 4 * (2 * cg - 1) : 2 * XCOPY + 8);
        ob1 = base + (cg <= 3 ? XCOPY + row * XROW + 8 * cg : 2 * XCOPY + 12);
    };
    int oa0, ob00, ob10, oa1, ob01, ob11;
    x_offsets(xrow0, xcg0, xq0, oa0, ob00, ob10);
    x_offsets(xrow1, xcg1, xq1, oa1, ob01, ob11);
    const int og = gq * GPITCH + (grow * QW + 4 * gcg) * 2;

    const int xoff0 = ((xrow0 * p.W + 4 * xcg0) * p.Cin + 4 * xq0) * 4, xoff1 = ((xrow1 * p.W + 4 * xcg1) * p.Cin + 4 * xq1) * 4;
    const int goff = ((grow * gW + 4 * gcg) * p.Cout + 4 * gq) * 4;
    const unsigned xbytes = (unsigned)(p.H * p.W * p.Cin) * 4u, gbytes = (unsigned)(gH * gW * p.Cout) * 4u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.styles ? p.styles : p.x), 0, p.styles ? (unsigned)(p.B * p.Cin) * 4u : 0u, 0x00020000);
    const float one = p.styles ? 0.f : 1.f;
    constexpr unsigned OOB = 0xfffffff0u;
    const size_t par_stride = (size_t)p.B * gH * gW * p.Cout;        // floats between parity images

    auto tile_origin = [&](int u, int& b, int& m0, int& n0) __attribute__((always_inline)) {
        const int tw = u % p.tiles_w, th = (u / p.tiles_w) % p.tiles_h;
        b = u / (p.tiles_w * p.tiles_h); m0 = th * QH; n0 = tw * QW;
    };
    // x patch of tile u: rows m0-1 .. m0+4, columns n0-1 .. n0+18 (5 column groups of 4)
    auto fetch_x = [&](int u) __attribute__((always_inline)) {
        const bool live = u < u_end;
        int b, m0, n0;
        tile_origin(live ? u : u_end - 1, b, m0, n0);
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.x + (size_t)b * p.H * p.W * p.Cin + ci0), 0, live ? xbytes - 4u * ci0 : 0u, 0x00020000);
        const int base_t = (((m0 - 1) * p.W + n0 - 1) * p.Cin) * 4;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int row = k ? xrow1 : xrow0, cg = k ? xcg1 : xcg0, q = k ? xq1 : xq0;
            // (shifts -1 / 0: five patch rows; a partial ci tile — Cin = 32 — loads zeros for the channels it does not have)
            const bool rowok = (unsigned)(m0 - 1 + row) < (unsigned)p.H && row < XR - 1 && ci0 + 4 * q < p.Cin;
            const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((b * p.Cin + ci0 + 4 * q) * 4), 0, 0));
            sx[k] = make_float4(v.x + one, v.y + one, v.z + one, v.w + one);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool ok = rowok && (unsigned)(n0 - 1 + 4 * cg + c) < (unsigned)p.W;
                const unsigned off = (unsigned)(base_t + c * p.Cin * 4 + (k ? xoff1 : xoff0));
                rx[k][c] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra, ok ? off : OOB, 0, 0));
            }
        }
    };
    // gradient tile of (tile u, parity PAR) into register set S
    auto fetch_g = [&](int u, auto par_tag, auto set_tag) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_tag)::value, S = decltype(set_tag)::value;
        const bool live = u < u_end;
        int b, m0, n0;
        tile_origin(live ? u : u_end - 1, b, m0, n0);
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.g + PAR * par_stride + (size_t)b * gH * gW * p.Cout + co0), 0, live ? gbytes - 4u * co0 : 0u, 0x00020000);
        const int base_t = ((m0 * gW + n0) * p.Cout) * 4;
        const bool rowok = m0 + grow < gH;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool ok = rowok && n0 + 4 * gcg + c < gW;
            const unsigned off = (unsigned)(base_t + c * p.Cout * 4 + goff);
            rg[S][c] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rb, ok ? off : OOB, 0, 0));
        }
    };
    // conversion pieces: x piece P = 0..7 (unit P >> 2, channel P & 3) into x stage `xs`; g piece e of set S into g stage `gs`
    auto x_piece = [&](int xs, auto piece_tag) __attribute__((always_inline)) {
        constexpr int P = decltype(piece_tag)::value, e = P & 3, k = P >> 2;
        const float* f0 = &rx[k][0].x; const float* f1 = &rx[k][1].x; const float* f2 = &rx[k][2].x; const float* f3 = &rx[k][3].x;
        const float sv = e == 0 ? sx[k].x : e == 1 ? sx[k].y : e == 2 ? sx[k].z : sx[k].w;
        uint2 hi, lo;
        split_run(f0[e] * sv, f1[e] * sv, f2[e] * sv, f3[e] * sv, hi, lo);
        char* row = lds + xs * UBX + 16 * e * XPITCH;
        *reinterpret_cast<uint2*>(row + (k ? oa1 : oa0)) = hi;
        *reinterpret_cast<uint2*>(row + (k ? oa1 : oa0) + XPART) = lo;
        *reinterpret_cast<unsigned*>(row + (k ? ob01 : ob00)) = hi.x;
        *reinterpret_cast<unsigned*>(row + (k ? ob11 : ob10)) = hi.y;
        *reinterpret_cast<unsigned*>(row + (k ? ob01 : ob00) + XPART) = lo.x;
        *reinterpret_cast<unsigned*>(row + (k ? ob11 : ob10) + XPART) = lo.y;
    };
    auto g_piece = [&](int gs, auto set_tag, auto e_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(set_tag)::value, e = decltype(e_tag)::value;
        const float* f0 = &rg[S][0].x; const float* f1 = &rg[S][1].x; const float* f2 = &rg[S][2].x; const float* f3 = &rg[S][3].x;
        uint2 hi, lo;
        split_run(f0[e], f1[e], f2[e], f3[e], hi, lo);
        char* dst = lds + 2 * UBX + gs * UBG + og + 16 * e * GPITCH;
        *reinterpret_cast<uint2*>(dst) = hi;
        *reinterpret_cast<uint2*>(dst + GPART) = lo;
    };

    const int abase = (32 * wi + l31) * XPITCH + 8 * h * 2;
    const int bbase = (32 * wj + l31) * GPITCH + 8 * h * 2;
    struct Raw { u32x4 ha, hb, la, lb; };
    auto read_a = [&](const char* st, int prow) __attribute__((always_inline)) {
        const char* ar = st + abase + prow * XROW;
        Raw r;
        r.ha = *reinterpret_cast<const u32x4*>(ar);
        r.hb = *reinterpret_cast<const u32x4*>(ar + XCOPY);
        r.la = *reinterpret_cast<const u32x4*>(ar + XPART);
        r.lb = *reinterpret_cast<const u32x4*>(ar + XPART + XCOPY);
        return r;
    };

    // one phase: K loop of (x stage xs, g stage J & 1) for parity PAR = (PP, PQ); riding along: the four g pieces of the NEXT
    // phase (register set (J + 1) & 1 -> g stage (J + 1) & 1) and, in the last two phases of a tile, four x pieces of the next tile
    auto phase = [&](int xs, auto j_tag) __attribute__((always_inline)) {
        constexpr int J = decltype(j_tag)::value;                      // phase of the tile: parities in the order 3, 1, 2, 0
        constexpr int PAR = J == 0 ? 3 : J == 1 ? 1 : J == 2 ? 2 : 0, PP = PAR >> 1, PQ = PAR & 1;
        constexpr int DYM = PP ? 2 : 3, DXM = PQ ? 2 : 3, NDY = popc3(DYM), NG = QH * NDY;
        constexpr int NP = J == 3 ? 12 : 4;                            // pieces that ride in this phase (the 4-tap phase also converts x)
        const char* xst = lds + xs * UBX;
        const char* gst = lds + 2 * UBX + (J & 1) * UBG;
        Raw cur_a = read_a(xst, nth3(DYM, 0));
        u32x4 bh = *reinterpret_cast<const u32x4*>(gst + bbase);
        u32x4 bl = *reinterpret_cast<const u32x4*>(gst + bbase + GPART);
        auto piece = [&](auto k_tag) __attribute__((always_inline)) {
            constexpr int K = decltype(k_tag)::value;
            if constexpr (K < 4) g_piece((J + 1) & 1, std::integral_constant<int, (J + 1) & 3>{}, std::integral_constant<int, (K < 4 ? K : 0)>{});
            else x_piece(xs ^ 1, std::integral_constant<int, (K >= 4 ? K - 4 : 0)>{});
        };
        auto step = [&](auto g_tag) __attribute__((always_inline)) {
            constexpr int G = decltype(g_tag)::value;
            constexpr int kr = G / NDY, dy = nth3(DYM, G % NDY);
            constexpr int G1 = G + 1, kr1 = G1 / NDY, dy1 = nth3(DYM, G1 % NDY);
            Raw nxt = cur_a;
            u32x4 nbh = bh, nbl = bl;
            if constexpr (G1 < NG) {
                nxt = read_a(xst, kr1 + dy1);
                if constexpr (kr1 != kr) {
                    nbh = *reinterpret_cast<const u32x4*>(gst + bbase + kr1 * QW * 2);
                    nbl = *reinterpret_cast<const u32x4*>(gst + bbase + GPART + kr1 * QW * 2);
                }
            }
            {
                u32x4 ah[2], al[2];                                    // windows dx = 0 (shift -1: copy A) and dx = 1 (shift 0)
                ah[0] = cur_a.ha; al[0] = cur_a.la;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ah[1][e] = __builtin_amdgcn_alignbit(cur_a.hb[e], cur_a.ha[e], 16);
                    al[1][e] = __builtin_amdgcn_alignbit(cur_a.lb[e], cur_a.la[e], 16);
                }
                constexpr int ky = PP + 2 * (1 - dy);                   // dy = 1: shift 0 -> sy = 0; dy = 0: shift -1 -> sy = 1
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)                          // products hi.hi, lo.hi, hi.lo
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx)
                        if ((DXM >> dx) & 1) {
                            const int t = 3 * ky + PQ + 2 * (1 - dx);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, pr == 1 ? al[dx] : ah[dx]),
                                                                             __builtin_bit_cast(bf16x8, pr == 2 ? bl : bh), acc[t], 0, 0, 0);
                        }
            }
            if constexpr (G < NP) piece(std::integral_constant<int, (G < NP ? G : 0)>{});
            if constexpr (G + NG < NP) piece(std::integral_constant<int, (G + NG < NP ? G + NG : 0)>{});
            __builtin_amdgcn_sched_barrier(0);
            cur_a = nxt; bh = nbh; bl = nbl;
        };
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{}); step(std::integral_constant<int, 3>{});
        if constexpr (NG > 4) {
            step(std::integral_constant<int, (NG > 4 ? 4 : 0)>{}); step(std::integral_constant<int, (NG > 4 ? 5 : 0)>{});
            step(std::integral_constant<int, (NG > 4 ? 6 : 0)>{}); step(std::integral_constant<int, (NG > 4 ? 7 : 0)>{});
        }
        __syncthreads();
    };

    // prologue: x of the first tile and the gradient tiles of its four phases (register set = phase); first x patch and first g tile
    // converted.  In the loop the gradient tile of (tile u+1, phase J) is loaded at the start of (tile u, phase J) — a whole tile
    // ahead: with two sets / two phases ahead (first version) a phase of 12 - 24 MFMAs was over before its successor's loads had
    // landed, SQ_WAIT_ANY 43 % of the wave cycles, MFMA pipe 0.27 busy — and converted inside the phase before its own.
    fetch_x(u_begin);
    fetch_g(u_begin, std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});
    fetch_g(u_begin, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
    fetch_g(u_begin, std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
    fetch_g(u_begin, std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
    x_piece(0, std::integral_constant<int, 0>{}); x_piece(0, std::integral_constant<int, 1>{});
    x_piece(0, std::integral_constant<int, 2>{}); x_piece(0, std::integral_constant<int, 3>{});
    x_piece(0, std::integral_constant<int, 4>{}); x_piece(0, std::integral_constant<int, 5>{});
    x_piece(0, std::integral_constant<int, 6>{}); x_piece(0, std::integral_constant<int, 7>{});
    g_piece(0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    g_piece(0, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    g_piece(0, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
    g_piece(0, std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
    __syncthreads();
    int xs = 0;
    for (int u = u_begin; u < u_end; ++u) {
        // (set J held tile u's phase-J tile, converted during phase J - 1: free when phase J starts)
        fetch_x(u + 1);                                                                          // converted inside phase 3
        fetch_g(u + 1, std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});
        phase(xs, std::integral_constant<int, 0>{});
        fetch_g(u + 1, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        phase(xs, std::integral_constant<int, 1>{});
        fetch_g(u + 1, std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
        phase(xs, std::integral_constant<int, 2>{});
        fetch_g(u + 1, std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
        phase(xs, std::integral_constant<int, 3>{});
        xs ^= 1;
    }
    store_slab_final(acc, lds, p.slabs + (size_t)ks * 9 * p.Cin * p.Cout, p.Cin, ci0, co0, wi, wj, h, l31, tid);
}

// host side: all nine taps of the up-sampling conv in one launch; slabs [ksplit][Cout][Cin][9], tap = 3 ky + kx
int launch_wgrad_up_bf16(const HfagpWgradArgs* a, hipStream_t s) {
    WgUpParams p{};
    p.x = a->x; p.g = a->g; p.styles = a->styles; p.slabs = a->workspace;
    p.B = a->B; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout; p.ksplit = a->ksplit;
    p.tiles_h = (a->H + 1 + QH - 1) / QH; p.tiles_w = (a->W + 1 + QW - 1) / QW;
    const size_t lds = (2 * UBX + 2 * UBG) > EPI_BYTES ? (size_t)(2 * UBX + 2 * UBG) : (size_t)EPI_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_up_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    wgrad_up_bf16_kernel<<<dim3((a->Cin + CT - 1) / CT, a->Cout / CT, a->ksplit), 256, lds, s>>>(p);
    return check_launch("conv_wgrad (up-sampling conv, split bf16)");
}

template <int DYM, int DXM, bool SWAP>
static int launch_wg16(const Wg16Params& p, dim3 grid, hipStream_t s) {
    const size_t lds = 2 * BUF > EPI_BYTES ? (size_t)(2 * BUF) : (size_t)EPI_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_bf16_kernel<DYM, DXM, SWAP>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    wgrad_bf16_kernel<DYM, DXM, SWAP><<<grid, 256, lds, s>>>(p);
    return check_launch("conv_wgrad (split bf16)");
}

// host side (called by hfagp_conv_wgrad when Cin, Cout are multiples of 64): the 3x3 conv ...
int launch_wgrad3x3_bf16(const HfagpWgradArgs* a, hipStream_t s) {
    Wg16Params p{};
    p.a = a->x; p.b = a->g; p.styles = a->styles; p.slabs = a->workspace;
    p.B = a->B; p.aH = a->H; p.aW = a->W; p.aC = a->Cin; p.bH = a->H; p.bW = a->W; p.bC = a->Cout;
    p.Cin = a->Cin; p.Cout = a->Cout; p.ksplit = a->ksplit;
    p.tiles_h = (a->H + QH - 1) / QH; p.tiles_w = (a->W + QW - 1) / QW;
    return launch_wg16<7, 7, false>(p, dim3(a->Cin / CT, a->Cout / CT, a->ksplit), s);
}

}  // namespace hfagp

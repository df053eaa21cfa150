// Weight gradient of the 3x3 modulated conv on v_mfma_f32_32x32x16_bf16 with split operands (gfx950):
//     dW[tap][ci][co] = sum_{b, pos} (x s)[b][pos + tap][ci] * g[b][pos][co]
// Same decomposition as wgrad.hip (M = ci, N = co, K = positions; a block owns a 64 x 64 (ci, co) tile for all nine
// taps and walks 2 x 16 position tiles; split-K slabs + wgrad_reduce_kernel), but the products are BF16X3: both
// operands are split into hi + lo bf16 parts (a gradient needs fp32's exponent range, which bf16 keeps) and a
// product is hi.hi + lo.hi + hi.lo accumulated in fp32 — 3 MFMAs of the 16x faster pipe instead of 8 fp32 MFMAs
// per 16 positions.
//
// A 16-bit MFMA operand is 8 K-CONTIGUOUS values per lane, and K = positions here, so the patches are TRANSPOSED on
// their way into LDS: x as [part][ci][patch position], g as [part][co][tile position] (a staging thread loads 4
// neighbouring positions x 4 channels and writes, per channel and part, one 8-byte run of 4 positions).  One MFMA
// K step = one row of 16 positions.  The tap shift dy picks the patch row; the shift dx moves the 8-position window
// by one element, which a 16-byte LDS read cannot do: the lane reads the aligned window plus one dword
// (patch columns 8h .. 8h+9) once per (row, part) and forms the three windows in registers — dx = -1: dwords 0-3,
// dx = +1: dwords 1-4, dx = 0: four v_alignbit_b32.
// LDS rows are PERMUTED: channel 4q + e of the tile lives in row 16 e + q, so that the 16 staging lanes of a column
// group (q = 0..15, one float4 = channels 4q..4q+3 each) write 16 consecutive rows (row pitch 52 / 20 dwords: 2-way
// instead of 8-way bank conflicts) and the 32 lanes of an MFMA operand read 32 consecutive rows (conflict-free);
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
    float* slabs;                       // [ksplit][ntaps][Cin][Cout], tap slot = row-major index over the (dy, dx) used
    int B, aH, aW, aC, bH, bW, bC;      // image extents / channel counts of the two operands (bH x bW = position grid)
    int Cin, Cout, tiles_h, tiles_w, ksplit;
};

constexpr int QH = 4, QW = 16;                   // position tile (4 rows: 108 MFMAs per wave hide one round of global loads)
constexpr int XR = QH + 2, XC = 24;              // x patch: 6 rows of 24 columns (18 used: halo 1 + 16 + 1)
constexpr int CT = 64;                           // ci / co tile
constexpr int XPITCH = XR * XC * 2 + 16;         // bytes per ci row of one part (208: 52 dwords, conflict-free b128)
constexpr int GPITCH = QH * QW * 2 + 16;         // bytes per co row of one part (80: 20 dwords)
constexpr int XPART = CT * XPITCH, GPART = CT * GPITCH;
constexpr int BUF = 2 * XPART + 2 * GPART;       // one stage: x hi, x lo, g hi, g lo  (57 344 B)

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

// DYM / DXM: bit i set = patch-row / window offset i is used (shift i - 1).  3x3: 7, 7 (nine taps).  Parity images of the
// up-conv: shifts {0, +1} = bits 1, 2 (6) or {0} (2).  SWAP: operand a is the gradient, b is x (styles go to b, and
// the tile is stored transposed into the [Cin][Cout] slab).
template <int DYM, int DXM, bool SWAP>
__global__ void __launch_bounds__(256, 1) wgrad_bf16_kernel(const Wg16Params p) {
    constexpr int NDX = popc3(DXM), NT = popc3(DYM) * NDX;
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
    // owns units t and 256 + t (t < 224); g: 4 rows x 4 column groups x 16 channel groups = 256 units, one per
    // thread.  Unit = 4 float4 loads (one per column, 4 channels each).
    // Loads are UNCONDITIONAL (out-of-image columns read element 0 of the image and are multiplied by a 0 mask at commit
    // time) and nothing in fetch() consumes a loaded value: the loads of tile u+1 stay in flight under the MFMAs of
    // tile u.  (With `if (inside) v = load; v *= style` in fetch the compiler waited for every load before the MFMAs.)
    float4 rx[2][4], rg[4], sx[2], sg = make_float4(1.f, 1.f, 1.f, 1.f);
    float mx[2][4], mg[4];
    auto unit_x = [&](int u, int& row, int& cg, int& q) { q = u & 15; cg = (u >> 4) % 5; row = (u >> 4) / 5; };
    auto fetch = [&](int u) {
        const int tw = u % p.tiles_w, th = (u / p.tiles_w) % p.tiles_h, b = u / (p.tiles_w * p.tiles_h);
        const int m0 = th * QH, n0 = tw * QW;
        const float* xb = p.a + (size_t)b * p.aH * p.aW * p.aC + ci0;
        const float* gb = p.b + (size_t)b * p.bH * p.bW * p.bC + co0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int uu = min(tid + 256 * k, XR * 5 * 16 - 1);           // threads past the last unit repeat it (same data, same address)
            int row, cg, q;
            unit_x(uu, row, cg, q);
            const int iy = m0 - 1 + row;
            const bool rowok = iy >= 0 && iy < p.aH;
            sx[k] = (p.styles && !SWAP) ? *reinterpret_cast<const float4*>(p.styles + (size_t)b * p.aC + ci0 + 4 * q)
                                        : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ix = n0 - 1 + 4 * cg + c;
                const bool ok = rowok && ix >= 0 && ix < p.aW;
                mx[k][c] = ok ? 1.f : 0.f;
                rx[k][c] = *reinterpret_cast<const float4*>(xb + (ok ? ((size_t)iy * p.aW + ix) * p.aC : 0) + 4 * q);
            }
        }
        {
            const int tg = tid;                                // QH * 4 * 16 = 256 units: one per thread
            const int q = tg & 15, cg = (tg >> 4) & 3, row = tg >> 6;
            const int iy = m0 + row;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int ix = n0 + 4 * cg + c;
                const bool ok = iy < p.bH && ix < p.bW;
                mg[c] = ok ? 1.f : 0.f;
                rg[c] = *reinterpret_cast<const float4*>(gb + (ok ? ((size_t)iy * p.bW + ix) * p.bC : 0) + 4 * q);
            }
            if constexpr (SWAP) sg = p.styles ? *reinterpret_cast<const float4*>(p.styles + (size_t)b * p.bC + co0 + 4 * q)
                                              : make_float4(1.f, 1.f, 1.f, 1.f);
        }
    };
    // commit in three pieces (x unit 0, x unit 1, g unit) so that the K loop can issue them between its MFMA groups
    auto commit_x = [&](int buf, auto k_tag) __attribute__((always_inline)) {
        constexpr int k = decltype(k_tag)::value;
        char* base = lds + buf * BUF;
        const int uu = min(tid + 256 * k, XR * 5 * 16 - 1);
        int row, cg, q;
        unit_x(uu, row, cg, q);
        const float sv[4] = {sx[k].x, sx[k].y, sx[k].z, sx[k].w};
        const float* f0 = &rx[k][0].x; const float* f1 = &rx[k][1].x;
        const float* f2 = &rx[k][2].x; const float* f3 = &rx[k][3].x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {             // channel 4q + e: columns 4cg .. 4cg+3
            uint2 hi, lo;
            split_run(f0[e] * (sv[e] * mx[k][0]), f1[e] * (sv[e] * mx[k][1]), f2[e] * (sv[e] * mx[k][2]),
                      f3[e] * (sv[e] * mx[k][3]), hi, lo);
            char* dst = base + (16 * e + q) * XPITCH + (row * XC + 4 * cg) * 2;
            *reinterpret_cast<uint2*>(dst) = hi;
            *reinterpret_cast<uint2*>(dst + XPART) = lo;
        }
    };
    auto commit_g = [&](int buf) __attribute__((always_inline)) {
        char* base = lds + buf * BUF;
        const int tg = tid;
        const int q = tg & 15, cg = (tg >> 4) & 3, row = tg >> 6;
        const float* f0 = &rg[0].x; const float* f1 = &rg[1].x; const float* f2 = &rg[2].x; const float* f3 = &rg[3].x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint2 hi, lo;
            const float se = e == 0 ? sg.x : e == 1 ? sg.y : e == 2 ? sg.z : sg.w;
            split_run(f0[e] * (se * mg[0]), f1[e] * (se * mg[1]), f2[e] * (se * mg[2]), f3[e] * (se * mg[3]), hi, lo);
            char* dst = base + 2 * XPART + (16 * e + q) * GPITCH + (row * QW + 4 * cg) * 2;
            *reinterpret_cast<uint2*>(dst) = hi;
            *reinterpret_cast<uint2*>(dst + GPART) = lo;
        }
    };
    auto commit = [&](int buf) {
        commit_x(buf, std::integral_constant<int, 0>{});
        commit_x(buf, std::integral_constant<int, 1>{});
        commit_g(buf);
    };

    // per-lane fragment bases (bytes inside a stage): A = LDS row 32 wi + l31, columns 8h..; B = LDS row 32 wj + l31
    const int abase = (32 * wi + l31) * XPITCH + 8 * h * 2;
    const int bbase = 2 * XPART + (32 * wj + l31) * GPITCH + 8 * h * 2;

    int cur = 0;
    if (u_begin < u_end) { fetch(u_begin); commit(0); }
    __syncthreads();
    for (int u = u_begin; u < u_end; ++u) {
        if (u + 1 < u_end) fetch(u + 1);
        const char* st = lds + cur * BUF;
        const bool more = u + 1 < u_end;
#pragma unroll
        for (int kr = 0; kr < QH; ++kr) {                 // K step = tile row kr (16 positions)
            u32x4 bh = *reinterpret_cast<const u32x4*>(st + bbase + kr * QW * 2);
            u32x4 bl = *reinterpret_cast<const u32x4*>(st + bbase + GPART + kr * QW * 2);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {              // patch row kr + dy  (dy - 1 = shift of operand a)
                if constexpr (((DYM >> 0) & 1) == 0) { if (dy == 0) continue; }
                if constexpr (((DYM >> 1) & 1) == 0) { if (dy == 1) continue; }
                if constexpr (((DYM >> 2) & 1) == 0) { if (dy == 2) continue; }
                const char* ar = st + abase + (kr + dy) * XC * 2;
                const u32x4 h4 = *reinterpret_cast<const u32x4*>(ar);
                const unsigned h5 = *reinterpret_cast<const unsigned*>(ar + 16);
                const u32x4 l4 = *reinterpret_cast<const u32x4*>(ar + XPART);
                const unsigned l5 = *reinterpret_cast<const unsigned*>(ar + XPART + 16);
                const unsigned hw[5] = {h4[0], h4[1], h4[2], h4[3], h5};
                const unsigned lw[5] = {l4[0], l4[1], l4[2], l4[3], l5};
                u32x4 ah[3], al[3];                       // the three dx windows (start column 8h + dx)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ah[0][e] = hw[e]; al[0][e] = lw[e];
                    ah[1][e] = __builtin_amdgcn_alignbit(hw[e + 1], hw[e], 16);
                    al[1][e] = __builtin_amdgcn_alignbit(lw[e + 1], lw[e], 16);
                    ah[2][e] = hw[e + 1]; al[2][e] = lw[e + 1];
                }
                const int trow = rank3(DYM, dy) * NDX;    // first tap slot of this row (dy is an unrolled constant)
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
            }
        }
        if (more) commit(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // ---- slab [tap slot][Cin][Cout]: C/D layout row = (r&3) + 8*(r>>2) + 4*h (operand-a channel), col = lane&31
    // (operand-b channel); SWAP: a = Cout side, b = Cin side
    float* slab = p.slabs + (size_t)ks * NT * p.Cin * p.Cout;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * h;                       // MFMA row -> channel (row permutation)
            const int ca = ci0 + 4 * (rr & 15) + 2 * wi + (rr >> 4), cb = co0 + 4 * (l31 & 15) + 2 * wj + (l31 >> 4);
            const int ci = SWAP ? cb : ca, co = SWAP ? ca : cb;
            slab[((size_t)t * p.Cin + ci) * p.Cout + co] = acc[t][r];
        }
}

template <int DYM, int DXM, bool SWAP>
static int launch_wg16(const Wg16Params& p, dim3 grid, hipStream_t s) {
    const size_t lds = 2 * BUF;
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

// ... and one parity image of the y_t gradient of the up-conv: g_par [B][H+1][W+1][Cout], taps = shifts
// {0, +1} x {0, +1} (parity 0), {0, +1} x {0} (1), {0} x {0, +1} (2), {0} x {0} (3); slabs [ksplit][ntaps][Cin][Cout]
int launch_wgrad_parity_bf16(const HfagpWgradArgs* a, const float* g_par, int parity, float* slabs, hipStream_t s) {
    Wg16Params p{};
    p.a = g_par; p.b = a->x; p.styles = a->styles; p.slabs = slabs;       // (slabs: this parity's region of the workspace)
    p.B = a->B; p.aH = a->H + 1; p.aW = a->W + 1; p.aC = a->Cout; p.bH = a->H; p.bW = a->W; p.bC = a->Cin;
    p.Cin = a->Cin; p.Cout = a->Cout; p.ksplit = a->ksplit;
    p.tiles_h = (a->H + QH - 1) / QH; p.tiles_w = (a->W + QW - 1) / QW;
    const dim3 grid(a->Cout / CT, a->Cin / CT, a->ksplit);
    switch (parity) {
        case 0: return launch_wg16<6, 6, true>(p, grid, s);
        case 1: return launch_wg16<6, 2, true>(p, grid, s);
        case 2: return launch_wg16<2, 6, true>(p, grid, s);
        default: return launch_wg16<2, 2, true>(p, grid, s);
    }
}

}  // namespace hfagp

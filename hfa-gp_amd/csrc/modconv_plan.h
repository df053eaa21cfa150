// Planning shared by the modulated-conv kernels (fp32-exact modconv.hip, split-bf16 modconv_bf16.hip): the
// phase/tap tables of the five conv modes, tiling, split-K choice.  gfx950 only.
#pragma once
#include <algorithm>
#include <cstdlib>
#include "common.h"

namespace hfagp {

constexpr int PW = 16;           // patch width in output positions
constexpr int MAXTAPS = 9;

struct Phase {
    int ntaps;
    int mh, mw;                  // extent of the (m, n) position grid of this phase
    int sy, sx, oy0, ox0;        // output pixel = (sy*m + oy0, sx*n + ox0)
    int slab;                    // output slab of this phase (phases that ACCUMULATE into the same pixels)
    long long in_off;            // element offset of this phase's input image inside x
    signed char dy[MAXTAPS], dx[MAXTAPS], widx[MAXTAPS];
};

struct ConvParams {
    const float* x; const void* wt; const float* styles; const float* dcoef;
    const float* noise; const float* bias;
    const float* x_absmax; float* y_absmax;    // fp16 range tracking (hfagp.h), may be null
    const float* rgb_w; float* rgb_part;       // fused toRGB (hfagp.h), may be null
    int x_f16, y_f16;                          // fp16 storage of x / y (hfagp.h)
    float* out;                  // y, or the split-K workspace
    long long x_batch_stride;
    long long slab;              // elements per split-K slab (B*Ho*Wo*Cout)
    int B, H, W, Cin, Cout, Ho, Wo;
    int in_h, in_w;              // extent of the input image(s) (differs from H, W for the parity images)
    int nphase, nslab, tiles_h, tiles_w, tiles_n, ksplit, nchunks;
    int wtaps;                   // taps of the weight image (9, or 1 for the 1x1 conv)
    int dymin, dxmin, ph, pw;    // patch origin offset and patch extent (pixels)
    int fused;                   // 1: apply the epilogue here, 0: store raw accumulators
    // merged up-conv (upconv_bf16_kernel) tiling: rows of all samples stacked with pitch up_rp (up_rows = B up_rp rows in up_tr tiles
    // of 8), up_tw column tiles of 16, up_nf fringe tiles for the column n = W (0: the regular tiles cover W+1 columns), styles /
    // range-guard scales of up_ns consecutive samples per block
    int up_rp, up_rows, up_tr, up_tw, up_nf, up_ns;
    int xcd;                     // 1: logical block id = xcd_remap(blockIdx.x) (contiguous tile ranges per XCD: siblings and neighbours share an L2)
    int act; float noise_strength, alpha, gain, clamp;
    Phase phase[4];
};

// ------------------------------------------------------------------ host side
struct Plan {
    ConvParams p;
    int bn;           // N tile: 128, 96, 64 or 32
    int bm;           // M tile: pixels per block
    dim3 grid;
    bool merged_up;   // split-bf16 CONVT3X3_UP2: one block computes all four output phases (upconv_bf16_kernel)
    bool merged_s2;   // split-bf16 CONVS2_BWD: one block runs the four parity phases into one accumulator set (no slabs)
    int up_waves;     // merged_up: 4 (N = 64 per block) or 8 waves (N = 128 per block, two waves per SIMD)
    size_t ws_bytes;
};

static inline void set_phase(Phase& ph, int ntaps, int mh, int mw, int sy, int sx, int oy0, int ox0,
                      const int (*taps)[3], int slab = 0, long long in_off = 0) {
    ph.ntaps = ntaps; ph.mh = mh; ph.mw = mw; ph.sy = sy; ph.sx = sx; ph.oy0 = oy0; ph.ox0 = ox0;
    ph.slab = slab; ph.in_off = in_off;
    for (int t = 0; t < ntaps; ++t) {
        ph.dy[t] = (signed char)taps[t][0]; ph.dx[t] = (signed char)taps[t][1]; ph.widx[t] = (signed char)taps[t][2];
    }
}

static inline int make_plan(const HfagpModconvArgs* a, Plan& pl, int ck) {
    ConvParams& p = pl.p;
    p = ConvParams{};
    p.x = a->x; p.wt = a->wt; p.styles = a->styles; p.dcoef = a->dcoef; p.noise = a->noise; p.bias = a->bias;
    p.x_absmax = a->x_absmax; p.y_absmax = a->y_absmax;
    p.rgb_w = a->rgb_w; p.rgb_part = a->rgb_part;
    p.x_f16 = a->x_f16; p.y_f16 = a->y_f16;
    p.x_batch_stride = a->x_batch_stride;
    p.B = a->B; p.H = a->H; p.W = a->W; p.Cin = a->Cin; p.Cout = a->Cout;
    p.act = a->act; p.noise_strength = a->noise_strength; p.alpha = a->alpha; p.gain = a->gain; p.clamp = a->clamp;
    p.nchunks = a->Cin / ck;
    { static const char* dev = getenv("HFAGP_DEV_XCD_REMAP"); p.xcd = dev ? atoi(dev) : 0; }
    p.nslab = 1;
    int in_h = a->H, in_w = a->W;      // extent of the image(s) the patches are read from

    // N tile / wave arrangement
    if (a->Cout % 128 == 0 || a->precision != HFAGP_PREC_F32) pl.bn = 128;     // 16-bit paths: 128-wide tiles only
    else if (a->Cout % 96 == 0) pl.bn = 96;
    else if (a->Cout % 64 == 0) pl.bn = 64;
    else pl.bn = 32;
    pl.bm = 128;
    const int PH = pl.bm / PW;
    p.tiles_n = (a->Cout + pl.bn - 1) / pl.bn;

    int gh, gw;   // extent of the position grid that the tiles cover
    if (a->mode == HFAGP_CONV3X3) {
        static const int t9[9][3] = {{-1, -1, 0}, {-1, 0, 1}, {-1, 1, 2}, {0, -1, 3}, {0, 0, 4},
                                     {0, 1, 5},   {1, -1, 6}, {1, 0, 7},  {1, 1, 8}};
        p.nphase = 1; p.Ho = a->H; p.Wo = a->W;
        set_phase(p.phase[0], 9, a->H, a->W, 1, 1, 0, 0, t9);
        p.dymin = -1; p.dxmin = -1; p.ph = PH + 2; p.pw = PW + 2;
        gh = a->H; gw = a->W; p.fused = 1;
    } else if (a->mode == HFAGP_CONV1X1) {
        static const int t1[1][3] = {{0, 0, 0}};
        p.nphase = 1; p.Ho = a->H; p.Wo = a->W;
        set_phase(p.phase[0], 1, a->H, a->W, 1, 1, 0, 0, t1);
        p.dymin = 0; p.dxmin = 0; p.ph = PH; p.pw = PW;
        gh = a->H; gw = a->W; p.fused = 1;
    } else if (a->mode == HFAGP_CONVT3X3_UP2) {
        // y_t[2i+ti][2j+tj] += x[i][j] * w[ti][tj]; output phase (a,b) collects ti = a (mod 2), tj = b (mod 2)
        static const int t00[4][3] = {{0, 0, 0}, {-1, 0, 6}, {0, -1, 2}, {-1, -1, 8}};
        static const int t01[2][3] = {{0, 0, 1}, {-1, 0, 7}};
        static const int t10[2][3] = {{0, 0, 3}, {0, -1, 5}};
        static const int t11[1][3] = {{0, 0, 4}};
        p.nphase = 4; p.Ho = 2 * a->H + 1; p.Wo = 2 * a->W + 1;
        set_phase(p.phase[0], 4, a->H + 1, a->W + 1, 2, 2, 0, 0, t00);
        set_phase(p.phase[1], 2, a->H + 1, a->W, 2, 2, 0, 1, t01);
        set_phase(p.phase[2], 2, a->H, a->W + 1, 2, 2, 1, 0, t10);
        set_phase(p.phase[3], 1, a->H, a->W, 2, 2, 1, 1, t11);
        p.dymin = -1; p.dxmin = -1; p.ph = PH + 1; p.pw = PW + 1;
        gh = a->H + 1; gw = a->W + 1; p.fused = 0;
    } else if (a->mode == HFAGP_CONV3X3_BWD) {
        // adjoint of mode 0 w.r.t. its input: dx[p][q] = sum_t g[p - dy_t][q - dx_t] . W_t^T  (taps mirrored)
        // (listed in ascending (dy, dx) like mode 0: modconv_bf16.hip relies on that order for its 9-tap loop)
        static const int t9[9][3] = {{-1, -1, 8}, {-1, 0, 7}, {-1, 1, 6}, {0, -1, 5}, {0, 0, 4},
                                     {0, 1, 3},   {1, -1, 2}, {1, 0, 1},  {1, 1, 0}};
        p.nphase = 1; p.Ho = a->H; p.Wo = a->W;
        set_phase(p.phase[0], 9, a->H, a->W, 1, 1, 0, 0, t9);
        p.dymin = -1; p.dxmin = -1; p.ph = PH + 2; p.pw = PW + 2;
        gh = a->H; gw = a->W; p.fused = 1;
    } else if (a->mode == HFAGP_CONVS2_BWD) {
        // adjoint of mode 1 w.r.t. its input: dx[i][j] = sum_{ti,tj} g_yt[2i+ti][2j+tj] . W_{ti,tj}^T.
        // x holds the four parity images of g_yt: x[a][b] [B][H+1][W+1][Cin], (a,b) = (ti&1, tj&1);
        // each parity is one phase accumulating into its own slab (summed by the split-K reducer).
        static const int t00[4][3] = {{0, 0, 0}, {1, 0, 6}, {0, 1, 2}, {1, 1, 8}};
        static const int t01[2][3] = {{0, 0, 1}, {1, 0, 7}};
        static const int t10[2][3] = {{0, 0, 3}, {0, 1, 5}};
        static const int t11[1][3] = {{0, 0, 4}};
        const long long img = (long long)a->B * (a->H + 1) * (a->W + 1) * a->Cin;
        p.nphase = 4; p.nslab = 4; p.Ho = a->H; p.Wo = a->W;
        set_phase(p.phase[0], 4, a->H, a->W, 1, 1, 0, 0, t00, 0, 0);
        set_phase(p.phase[1], 2, a->H, a->W, 1, 1, 0, 0, t01, 1, img);
        set_phase(p.phase[2], 2, a->H, a->W, 1, 1, 0, 0, t10, 2, 2 * img);
        set_phase(p.phase[3], 1, a->H, a->W, 1, 1, 0, 0, t11, 3, 3 * img);
        p.dymin = 0; p.dxmin = 0; p.ph = PH + 1; p.pw = PW + 1;
        in_h = a->H + 1; in_w = a->W + 1;
        gh = a->H; gw = a->W; p.fused = 0;
    } else {
        set_error("modconv: unknown mode %d", a->mode);
        return HFAGP_EBADARG;
    }
    p.in_h = in_h; p.in_w = in_w;
    p.wtaps = a->mode == HFAGP_CONV1X1 ? 1 : 9;
    p.tiles_h = (gh + PH - 1) / PH;
    p.tiles_w = (gw + PW - 1) / PW;
    p.slab = (long long)a->B * p.Ho * p.Wo * a->Cout;

    pl.merged_up = a->precision != HFAGP_PREC_F32 && a->mode == HFAGP_CONVT3X3_UP2;
    pl.merged_s2 = a->precision != HFAGP_PREC_F32 && a->mode == HFAGP_CONVS2_BWD;
    if (pl.merged_s2) {                // the parity phases accumulate in registers: one slab (hfagp_modconv_workspace_bytes follows)
        p.nslab = 1;
        for (int q = 0; q < 4; ++q) p.phase[q].slab = 0;
    }
    long long up_tiles = 0;
    if (pl.merged_up) {
        // see upconv_bf16_kernel: stacked rows, exact column tiles + fringe tiles.  A run of L consecutive stacked rows touches at
        // most (L + RP - 2) / RP + 1 samples: 9 rows for a regular tile's patch, 65 for a fringe tile's.
        const int xb = a->x_f16 ? 2 : 4;
        auto span = [](int L, int rp) { return (L + rp - 2) / rp + 1; };
        auto fits = [&](int ns) { return ns <= 8 && (ns <= 1 || (long long)ns * a->x_batch_stride * xb < (1ll << 32)); };
        int rp = a->H + 1;
        { static const char* dev = getenv("HFAGP_DEV_UP_LEGACY_TILES"); if (dev && atoi(dev) == 1) rp = -1; }
        bool fr = rp > 0 && a->W % PW == 0;
        { static const char* dev = getenv("HFAGP_DEV_UP_NO_FRINGE"); if (dev && atoi(dev) == 1) fr = false; }
        int ns = rp > 0 ? std::min(a->B, std::max(span(PH + 1, rp), fr ? span(8 * PH + 1, rp) : 1)) : 99;
        if (!fits(ns) && fr) { fr = false; ns = std::min(a->B, span(PH + 1, rp)); }
        if (!fits(ns)) { rp = (a->H + 1 + PH - 1) / PH * PH; fr = false; ns = std::min(a->B, 2); }   // per-sample row tiles (rounds 2-5)
        p.up_rp = rp; p.up_rows = a->B * rp; p.up_ns = ns;
        p.up_tr = (p.up_rows + PH - 1) / PH;
        p.up_tw = fr ? a->W / PW : (a->W + 1 + PW - 1) / PW;
        p.up_nf = fr ? (p.up_rows + 8 * PH - 1) / (8 * PH) : 0;
        up_tiles = (long long)p.up_tr * p.up_tw + p.up_nf;
    }
    // Round 6: FOUR waves (N = 64 per block) at <= 256 registers, so that TWO independent blocks share a CU — their prologues (styles,
    // range guard, first patch from HBM) and store epilogues overlap the other block's K loop.  The 8-wave block (N = 128, patch staged
    // once for twice the MFMA work, but one block per CU with all eight waves in lockstep) measured 1-16 % slower at every size and
    // batch (profiles/r06_up_waves_ab.log); it stays for bf16x6 (three parts: the 4-wave kernel only fits one block per CU there,
    // where the wider block at least halves the staging) behind the developer switch below.
    pl.up_waves = 4;
    (void)up_tiles;
    { static const char* dev = getenv("HFAGP_DEV_UP_WAVES"); if (dev && pl.merged_up) pl.up_waves = atoi(dev) == 8 && a->Cout % 128 == 0 ? 8 : 4; }
    const int grid_tiles_n = pl.merged_up ? a->Cout / (pl.up_waves == 8 ? 128 : 64) : p.tiles_n;
    const int grid_phases = (pl.merged_up || pl.merged_s2) ? 1 : p.nphase;
    const long long base_blocks = pl.merged_up ? up_tiles * grid_tiles_n
                                               : (long long)p.tiles_h * p.tiles_w * a->B * grid_tiles_n * grid_phases;
    int ks = a->ksplit;
    if (ks <= 0) {
        // split K until every CU has ONE block, and no further (round 4 sweep, tools/dev/ksplit_sweep.py: the old target of two
        // blocks per CU, rounded up, over-split the 32^2 ... 64^2 layers at batch 1 - 2 — every extra slab is another pass over
        // the output for the GEMM and for the reducer: 512 -> 512 up-conv to 64^2 at B = 2 59 -> 51 us with ks 3 -> 1, B = 1
        // 59 -> 38 us with ks 5 -> 2; 3x3 at 32^2 B = 1 41 -> 36 us with ks 16 -> 8)
        ks = 1;
        const long long target = kNumCU;
        if (base_blocks < target) ks = (int)(target / base_blocks);
        if (ks < 1) ks = 1;
        const int max_ks = p.nchunks / 2 > 0 ? p.nchunks / 2 : 1;   // at least 2 chunks per split
        if (ks > max_ks) ks = max_ks;
        if (ks > 64) ks = 64;
    }
    if (ks > p.nchunks) ks = p.nchunks;
    p.ksplit = ks;
    if (ks * p.nslab > 1) p.fused = 0;
    pl.ws_bytes = ks * p.nslab > 1 ? (size_t)ks * p.nslab * p.slab * sizeof(float) : 0;
    pl.grid = dim3((unsigned)(base_blocks / grid_phases * ks), (unsigned)grid_phases, 1);
    return HFAGP_OK;
}

static inline int validate(const HfagpModconvArgs* a, int ck) {
    // (y may be NULL when the fused toRGB sums are the only output wanted: hfagp.h, rgb_part)
    HFAGP_REQUIRE(a && a->x && a->wt && (a->y || a->rgb_part), HFAGP_EBADARG, "modconv: null pointer");
    HFAGP_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->Cin > 0 && a->Cout > 0, HFAGP_EBADARG, "modconv: bad dims");
    HFAGP_REQUIRE(a->Cin % ck == 0, HFAGP_EUNSUPPORTED, "modconv: Cin=%d must be a multiple of %d", a->Cin, ck);
    HFAGP_REQUIRE(a->Cout % 4 == 0, HFAGP_EUNSUPPORTED, "modconv: Cout=%d must be a multiple of 4", a->Cout);
    HFAGP_REQUIRE(a->act == HFAGP_ACT_LINEAR || a->act == HFAGP_ACT_LRELU, HFAGP_EUNSUPPORTED, "modconv: act %d", a->act);
    return HFAGP_OK;
}


// modconv_bf16.hip
int launch_modconv_bf16(const HfagpModconvArgs* a, Plan& pl, hipStream_t s);

// smallconv.hip: the small-image kernel (lean 32 x 32 blocks without staging; K sliced over blocks through the workspace and the
// split-K reducer like the staged kernel, smallconv_ksplit).  Taken for the
// 3x3 conv, its data adjoint (images of at most 256 positions: 4^2 ... 16^2) and the 1x1 conv (at most 1024: ... 32^2) on 16-bit
// weight images when the caller did not ask for a particular split (ksplit <= 0).
int launch_smallconv(const HfagpModconvArgs* a, Plan& pl, hipStream_t s);
int smallconv_ksplit(const HfagpModconvArgs* a);                 // K slices (1: epilogue in the kernel, no workspace)
static inline bool smallconv_takes(const HfagpModconvArgs* a) {
    return a->precision != HFAGP_PREC_F32 && a->ksplit <= 0 &&
           (a->mode == HFAGP_CONV3X3 || a->mode == HFAGP_CONV1X1 || a->mode == HFAGP_CONV3X3_BWD) &&
           // (3x3 at 32^2 measured 58 us here against 33 + 6 us for the staged kernel: 9 taps re-read the activations from L2 nine
           // times; the 1x1 has no such re-read: 7.5 + 5.5 us against 22 + 6 us at 32^2)
           (long long)a->H * a->W <= (a->mode == HFAGP_CONV1X1 ? 1024 : 256) &&
           // (round 6, tools/dev/smallconv_ab.sh -> profiles/r06_smallconv_ab.log: the lean kernel is a SMALL-BATCH kernel.  With >= 1024
           // positions in the batch the staged kernel at its own split-K choice wins on the 3x3 layers of 8^2 and 16^2: B = 32: 433 ->
           // 122 us at 16^2, 111 -> 62 us at 8^2; B = 8 at 16^2: 113 -> 50 us; at 4^2 the lean kernel wins at every batch)
           !(a->mode != HFAGP_CONV1X1 && (long long)a->H * a->W >= 64 && (long long)a->B * a->H * a->W >= 1024) &&
           a->Cin % 16 == 0 && a->Cin <= 512 && a->Cout % 32 == 0 && !a->x_f16 && !a->y_f16 &&
           !a->rgb_w && !a->rgb_part && a->y != nullptr;
}

}  // namespace hfagp

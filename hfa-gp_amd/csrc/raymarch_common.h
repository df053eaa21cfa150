// Device code shared by the forward ray marcher (raymarch.hip) and its backward pass (raymarch_bwd.hip):
// ray generation, tri-plane bilinear gather in the MFMA B-operand layout, and the decoder MLP on
// v_mfma_f32_16x16x4_f32.  See raymarch.hip for the mapping of lanes to samples / channels.
#pragma once
#include "common.h"

namespace hfagp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CS = 36;   // LDS colour row stride in floats (32 + pad: conflict-free b128 writes)

struct RayParams {
    HfagpRaymarchArgs a;
    float lin_step;      // (float(end) - float(start)) / (Sc - 1)   [torch.linspace, fp32]
    float delta;         // float( (end - start) / (Sc - 1) )        [python double -> fp32]
    float coord_scale;   // 2 / box_warp
    int total_rays;
    const float* g_feat; // backward only: [B][R][32]
    float* rec;          // backward only: per-sample records [B*R][S][4] = (depth, omega, d sigma, -)
};

inline int fill_ray_params(const HfagpRaymarchArgs* a, RayParams& p, const char* who) {
    HFAGP_REQUIRE(a->planes && a->cam2world && a->intrinsics && a->u_strat && a->u_imp && a->dec_w0 && a->dec_b0 &&
                      a->dec_w1 && a->dec_b1,
                  HFAGP_EBADARG, "%s: null pointer", who);
    HFAGP_REQUIRE(a->B > 0 && a->H > 1 && a->W > 1 && a->res > 0, HFAGP_EBADARG, "%s: bad dims", who);
    HFAGP_REQUIRE(a->box_warp > 0.f && a->ray_end > a->ray_start, HFAGP_EBADARG, "%s: bad ray range", who);
    HFAGP_REQUIRE(a->Sc == a->Sf && (a->Sc == 16 || a->Sc == 32 || a->Sc == 48), HFAGP_EUNSUPPORTED,
                  "%s: unsupported sample counts Sc=%d Sf=%d (supported: 16+16, 32+32, 48+48)", who, a->Sc, a->Sf);
    p.a = *a;
    p.lin_step = ((float)a->ray_end - (float)a->ray_start) / (float)(a->Sc - 1);
    p.delta = (float)((a->ray_end - a->ray_start) / (double)(a->Sc - 1));
    p.coord_scale = (float)(2.0 / (double)a->box_warp);
    const long long total = (long long)a->B * a->res * a->res;
    HFAGP_REQUIRE(total < (1ll << 31), HFAGP_EUNSUPPORTED, "%s: too many rays", who);
    p.total_rays = (int)total;
    p.g_feat = nullptr;
    p.rec = nullptr;
    return HFAGP_OK;
}

int launch_raymarch(const RayParams& p, bool grads, hipStream_t s);   // raymarch.hip

// sort + gather form of the backward pass 2 (raymarch_rows.hip): where the dL/dF producers write a sample's line
struct RowsOut {
    const int2* pos;     // [samples][P] (slot, slot of the second entry or -1)
    unsigned* dfs;       // [slots][32] bf16 hi << 16 | bf16 lo
    int P;               // planes scattered (2: planes 1 and 2 mirror each other)
};
size_t rows_scratch_bytes(const HfagpRaymarchArgs& a);                                         // 0: variant does not apply
int rows_prepare(const RayParams& p, void* scratch, size_t bytes, RowsOut& out, hipStream_t s);    // S1, scan, S2
int rows_gather(const RayParams& p, void* scratch, float* d_planes, hipStream_t s);               // G

// XCD-local ray schedule (MI355X: 8 XCDs, each with its own 4 MB L2; block b runs on XCD b % 8).  The planes of a frame
// are 25 MB — no L2 holds them — and with rays dealt round-robin over the blocks every XCD walks every region of every
// frame's planes: measured 2.07 GB of L2 fills per 8-frame launch against 0.22 GB compulsory.  Here
//   * the ray sequence is cut into 8 contiguous parts, one per XCD (8 frames: one frame per XCD; 1 frame: an eighth);
//   * inside a frame the sequence runs over 16-pixel-wide COLUMN strips, row by row: rays of one strip share their
//     x range, so the (x,z) and (z,x) planes they touch are two 32-texel-wide bands (1 MB each) that stay in the XCD's
//     L2 for the whole strip while the (x,y) plane streams through once.
// The mapping is a bijection of the ray indices whatever the real block placement is (placement only affects speed).
struct RaySchedule {
    long long begin, end; int stride;        // this wave's positions in the sequence: begin, begin + stride, ... < end
};

__device__ __forceinline__ RaySchedule ray_schedule(long long total, int wave, int waves_per_block = 4) {
    const unsigned g = gridDim.x;
    const unsigned l = xcd_remap(blockIdx.x, g);                 // logical block id: contiguous per XCD
    const unsigned q = g / kNumXCD, r = g % kNumXCD;              // blocks per XCD: q + 1 for the first r XCDs, q for the rest
    unsigned xcd, local, nloc;
    if (l < r * (q + 1)) { xcd = l / (q + 1); local = l % (q + 1); nloc = q + 1; }
    else { xcd = r + (l - r * (q + 1)) / max(q, 1u); local = (l - r * (q + 1)) % max(q, 1u); nloc = q; }
    if (g < (unsigned)kNumXCD) { xcd = l; local = 0; nloc = 1; }  // fewer blocks than XCDs: one part per block ...
    const unsigned parts = g < (unsigned)kNumXCD ? g : (unsigned)kNumXCD;
    const long long per = (total + parts - 1) / parts;
    RaySchedule s;
    s.begin = min(total, xcd * per + (long long)local * waves_per_block + wave);
    s.end = min(total, (xcd + 1) * per);
    s.stride = (int)nloc * waves_per_block;
    return s;
}

// position in the sequence -> (frame, pixel row, pixel column)
__device__ __forceinline__ void ray_of(int i, int res, int& b, int& pi, int& pj) {
    const int R = res * res;
    b = i / R;
    const int rr = i % R;
    constexpr int SW = 16;
    if (res % SW == 0) {
        const int strip = rr / (SW * res), within = rr % (SW * res);
        pi = within / SW;
        pj = strip * SW + within % SW;
    } else {
        pi = rr / res;
        pj = rr % res;
    }
}

// Waves of a workgroup are independent here; LDS hand-offs between lanes of ONE wave only need the
// compiler not to reorder the accesses (the LDS executes a wave's DS instructions in order).
#define WAVE_SYNC()                                            \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
    } while (0)

// Transcendentals on the hardware units (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp each).  The libm
// forms (expf / log1pf / IEEE division) cost ~55 VALU instructions per softplus and made the kernel
// VALU-bound (17 k VALU instructions per ray, rocprofv3 SQ_INSTS_VALU); these cost ~8.
//   softplus(x) = max(x, 0) + log(1 + exp(-|x|))   (argument of log in (1, 2]: abs error ~1e-7;
//                                                   equals x for x > 20 like torch's threshold form)
__device__ __forceinline__ float exp_f(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float log_f(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + log_f(1.f + exp_f(-fabsf(x))); }
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + exp_f(-x)); }

__device__ __forceinline__ float wave_scan_mul(float v, int lane) {   // inclusive product scan, 64 lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float u = __shfl_up(v, o);
        if (lane >= o) v *= u;
    }
    return v;
}
__device__ __forceinline__ float wave_scan_add(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------- ray generation (RaySampler.forward)
__device__ __forceinline__ void ray_setup(const HfagpRaymarchArgs& a, int b, int pi, int pj, float o3[3], float d3[3]) {
    const float* M = a.cam2world + b * 16;
    const float* K = a.intrinsics + b * 9;
    const float fx = K[0], sk = K[1], cx = K[2], fy = K[4], cy = K[5];
    const float inv_res = 1.0f / (float)a.res, half_res = 0.5f / (float)a.res;
    const float xc = __fadd_rn(__fmul_rn((float)pj, inv_res), half_res);
    const float yc = __fadd_rn(__fmul_rn((float)pi, inv_res), half_res);
    const float xl = __fdiv_rn(__fsub_rn(__fadd_rn(__fsub_rn(xc, cx), __fdiv_rn(__fmul_rn(cy, sk), fy)),
                                         __fdiv_rn(__fmul_rn(sk, yc), fy)), fx);
    const float yl = __fdiv_rn(__fsub_rn(yc, cy), fy);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float wv = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(M[4 * k], xl), __fmul_rn(M[4 * k + 1], yl)),
                                             M[4 * k + 2]), M[4 * k + 3]);
        o3[k] = M[4 * k + 3];
        d3[k] = __fsub_rn(wv, o3[k]);
    }
    const float nrm = fmaxf(__fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(d3[0], d3[0]), __fmul_rn(d3[1], d3[1])),
                                                 __fmul_rn(d3[2], d3[2]))), 1e-12f);
    d3[0] = __fdiv_rn(d3[0], nrm); d3[1] = __fdiv_rn(d3[1], nrm); d3[2] = __fdiv_rn(d3[2], nrm);
}

// ---------------------------------------------------------------- tri-plane gather
struct PlaneTaps {            // the 4 bilinear taps of one plane: texel index (y*W + x) and weight (0 if outside)
    int idx[4];
    float w[4];
};

// F.grid_sample(bilinear, zeros, align_corners=False) as ATen's CPU kernel computes it:
// pixel = (g + 1) * (size / 2) - 0.5;  weights nw = (1-fy)(1-fx), ne = (1-fy)fx, sw = fy(1-fx), se = fy fx
__device__ __forceinline__ void plane_pixel(const HfagpRaymarchArgs& a, float gx, float gy, float& ix, float& iy) {
    ix = __fsub_rn(__fmul_rn(__fadd_rn(gx, 1.f), (float)a.W * 0.5f), 0.5f);
    iy = __fsub_rn(__fmul_rn(__fadd_rn(gy, 1.f), (float)a.H * 0.5f), 0.5f);
}

__device__ __forceinline__ void plane_taps(const HfagpRaymarchArgs& a, float gx, float gy, PlaneTaps& t) {
    const float fW = (float)a.W, fH = (float)a.H;
    float ix, iy;
    plane_pixel(a, gx, gy, ix, iy);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const float we = __fsub_rn(ix, fx0), ww = __fsub_rn(1.f, we);
    const float ws_ = __fsub_rn(iy, fy0), wn = __fsub_rn(1.f, ws_);
    // clamp before the int conversion so far-away coordinates stay defined
    const int x0 = (int)fminf(fmaxf(fx0, -2.f), fW + 1.f), y0 = (int)fminf(fmaxf(fy0, -2.f), fH + 1.f);
    const int x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < a.W, vx1 = x1 >= 0 && x1 < a.W;
    const bool vy0 = y0 >= 0 && y0 < a.H, vy1 = y1 >= 0 && y1 < a.H;
    const int cx0 = min(max(x0, 0), a.W - 1), cx1 = min(max(x1, 0), a.W - 1);
    const int cy0 = min(max(y0, 0), a.H - 1), cy1 = min(max(y1, 0), a.H - 1);
    t.w[0] = (vx0 && vy0) ? __fmul_rn(wn, ww) : 0.f;
    t.w[1] = (vx1 && vy0) ? __fmul_rn(wn, we) : 0.f;
    t.w[2] = (vx0 && vy1) ? __fmul_rn(ws_, ww) : 0.f;
    t.w[3] = (vx1 && vy1) ? __fmul_rn(ws_, we) : 0.f;
    t.idx[0] = cy0 * a.W + cx0; t.idx[1] = cy0 * a.W + cx1;
    t.idx[2] = cy1 * a.W + cx0; t.idx[3] = cy1 * a.W + cx1;
}

// sample position -> the three plane projections: (x,y), (x,z), (z,x) [eg3d original] or (z,y) [fixed]
__device__ __forceinline__ void sample_point(const RayParams& p, const float o3[3], const float d3[3], float tz, float q[3]) {
#pragma unroll
    for (int k = 0; k < 3; ++k) q[k] = __fmul_rn(p.coord_scale, __fadd_rn(o3[k], __fmul_rn(tz, d3[k])));
}
// grid coordinates (gx, gy) of plane pl for the normalised sample point q
__device__ __forceinline__ void plane_coords(const HfagpRaymarchArgs& a, const float q[3], int pl, float& gx, float& gy) {
    gx = pl == 2 ? q[2] : q[0];
    gy = pl == 0 ? q[1] : pl == 1 ? q[2] : (a.plane_axes == 0 ? q[0] : q[1]);
}

__device__ __forceinline__ void sample_taps(const RayParams& p, const float o3[3], const float d3[3], float tz,
                                            PlaneTaps taps[3]) {
    float q[3];
    sample_point(p, o3, d3, tz, q);
    plane_taps(p.a, q[0], q[1], taps[0]);
    plane_taps(p.a, q[0], q[2], taps[1]);
    plane_taps(p.a, q[2], p.a.plane_axes == 0 ? q[0] : q[1], taps[2]);
}

// lane (j, g) accumulates channels 8g..8g+7 of the mean over the 3 planes of the bilinear samples
__device__ __forceinline__ void gather8(const HfagpRaymarchArgs& a, int b, int g, const PlaneTaps taps[3], float f[8]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] = 0.f;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        // uniform plane base (b is wave-uniform: callers pass it through readfirstlane) + 32-bit lane offset
        const char* base = reinterpret_cast<const char*>(a.planes + ((size_t)(b * 3 + pl) * a.H * a.W) * 32);
        float4 v0[4], v1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned off = ((unsigned)taps[pl].idx[k] * 32u + 8u * g) * 4u;      // < 2^32: one plane
            v0[k] = *reinterpret_cast<const float4*>(base + off);
            v1[k] = *reinterpret_cast<const float4*>(base + off + 16);
        }
        const float v[4][8] = {{v0[0].x, v0[0].y, v0[0].z, v0[0].w, v1[0].x, v1[0].y, v1[0].z, v1[0].w},
                               {v0[1].x, v0[1].y, v0[1].z, v0[1].w, v1[1].x, v1[1].y, v1[1].z, v1[1].w},
                               {v0[2].x, v0[2].y, v0[2].z, v0[2].w, v1[2].x, v1[2].y, v1[2].z, v1[2].w},
                               {v0[3].x, v0[3].y, v0[3].z, v0[3].w, v1[3].x, v1[3].y, v1[3].z, v1[3].w}};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float acc = v[0][c] * taps[pl].w[0];
            acc = fmaf(v[1][c], taps[pl].w[1], acc);
            acc = fmaf(v[2][c], taps[pl].w[2], acc);
            acc = fmaf(v[3][c], taps[pl].w[3], acc);
            f[c] += acc;
        }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] *= 0.3333333333333333f;   // mean over the 3 planes
}

// ---------------------------------------------------------------- decoder (OSGDecoder) on the matrix core
// Effective weights W * lr_mul / sqrt(fan_in) as MFMA A-operand registers of lane (j = lane&15, g = lane>>4):
//   w0a[mt][t]      = W0[16mt + j][8g + t]            layer 1, K step t uses channel 8g + t
//   w1a[ot][4mt+r]  = W1[1 + 16ot + j][16mt + 4g + r] layer 2, K step (mt, r) uses hidden 16mt + 4g + r
//   wsig[mt][r]     = W1[0][16mt + 4g + r]            sigma row, VALU
struct DecoderRegs {
    float w0a[4][8], b0c[4][4], wsig[4][4], w1a[2][16], b1c[2][4];
    float bsig;
};

__device__ __forceinline__ void load_decoder(const HfagpRaymarchArgs& a, int j, int g, DecoderRegs& w) {
    const float g0 = a.decoder_lr_mul * 0.17677669529663687f;   // 1/sqrt(32)
    const float g1 = a.decoder_lr_mul * 0.125f;                 // 1/sqrt(64)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int t = 0; t < 8; ++t) w.w0a[mt][t] = a.dec_w0[(16 * mt + j) * 32 + 8 * g + t] * g0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            w.b0c[mt][r] = a.dec_b0[16 * mt + 4 * g + r] * a.decoder_lr_mul;
            w.wsig[mt][r] = a.dec_w1[16 * mt + 4 * g + r] * g1;
        }
    }
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                w.w1a[ot][mt * 4 + r] = a.dec_w1[(1 + 16 * ot + j) * 64 + 16 * mt + 4 * g + r] * g1;
#pragma unroll
        for (int r = 0; r < 4; ++r) w.b1c[ot][r] = a.dec_b1[1 + 16 * ot + 4 * g + r] * a.decoder_lr_mul;
    }
    w.bsig = a.dec_b1[0] * a.decoder_lr_mul;
}

// f[8] (B operand) -> hidden pre-activations hp, softplus values h (C layout: lane (j,g), rows 4g+r of tile mt),
// sigma (all four g lanes of a sample hold it), colour logits o[2] (rows 4g+r of colour tile ot).
template <bool KEEP_PRE>
__device__ __forceinline__ void decoder_fwd(const DecoderRegs& w, const float f[8], f32x4 hp[4], f32x4 h[4],
                                            float& sigma, f32x4 o[2]) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        h[mt] = f32x4{w.b0c[mt][0], w.b0c[mt][1], w.b0c[mt][2], w.b0c[mt][3]};
#pragma unroll
        for (int t = 0; t < 8; ++t) h[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w0a[mt][t], f[t], h[mt], 0, 0, 0);
    }
    float sg = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (KEEP_PRE) hp[mt][r] = h[mt][r];
            h[mt][r] = softplus_f(h[mt][r]);
            sg = fmaf(h[mt][r], w.wsig[mt][r], sg);
        }
    sg += __shfl_xor(sg, 16);
    sg += __shfl_xor(sg, 32);
    sigma = sg + w.bsig;
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
        o[ot] = f32x4{w.b1c[ot][0], w.b1c[ot][1], w.b1c[ot][2], w.b1c[ot][3]};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w1a[ot][mt * 4 + r], h[mt][r], o[ot], 0, 0, 0);
            }
    }
}

// ---------------------------------------------------------------- the decoder on the 16-bit matrix pipe
// v_mfma_f32_16x16x32_f16 with SPLIT operands (fp16 hi + lo parts, products hi.hi + lo.hi + hi.lo: ~2^-22, the F16X3
// arithmetic of the conv kernels): layer 1 is ONE K step of 32 features (4 M tiles x 3 products = 12 MFMAs instead of
// 32 fp32 ones at twice the issue time each), layer 2 two K steps of 32 hidden units (12 instead of 32).  The operand
// layouts are those of the fp32 version: B = the lane's 8 gathered channels k = 8g + c; the C registers of layer 1
// (rows 16mt + 4g + r) are the B operand of layer 2 with K step ks <-> tiles mt = 2ks, 2ks + 1 (k = 4 (mt & 1) + r) and
// the weight columns permuted to match.
// fp16 has 5 exponent bits, so every operand is scaled by an exact power of two into [.., 2^14]: the features by the
// caller's bound on |planes| (HfagpRaymarchArgs::planes_absmax), the hidden units by max_i (|b0_i| + sum_k |W0_ik| M),
// the weights by their own maxima; the accumulators are scaled back with one multiply.  Results equal the un-scaled
// arithmetic; values more than 2^18 below their tensor's bound lose low bits only (absolute error < bound * 2^-40).
typedef _Float16 f16x8r __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8r __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2r __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2r __attribute__((ext_vector_type(2)));
typedef float f32x2r __attribute__((ext_vector_type(2)));
typedef unsigned u32x4r __attribute__((ext_vector_type(4)));

// 8 floats -> hi / lo fp16 parts (each part: 4 dwords of 2 halves), lo = fp16(v - float(hi))
__device__ __forceinline__ void split8_f16(const float v[8], u32x4r& hi, u32x4r& lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x2r x = {v[2 * q], v[2 * q + 1]};
        const f16x2r hh = __builtin_convertvector(x, f16x2r);
        const f32x2r r = {v[2 * q] - (float)hh[0], v[2 * q + 1] - (float)hh[1]};
        hi[q] = __builtin_bit_cast(unsigned, hh);
        lo[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2r));
    }
}
// the same with bfloat16 parts (gradients: a raw gradient needs fp32's exponent range; ~2^-16 per product)
__device__ __forceinline__ void split8_bf16(const float v[8], u32x4r& hi, u32x4r& lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x2r x = {v[2 * q], v[2 * q + 1]};
        const unsigned hh = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2r));
        const f32x2r r = {v[2 * q] - __builtin_bit_cast(float, hh << 16), v[2 * q + 1] - __builtin_bit_cast(float, hh & 0xffff0000u)};
        hi[q] = hh;
        lo[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2r));
    }
}
__device__ __forceinline__ f32x4 mfma3_f16(u32x4r ah, u32x4r al, u32x4r bh, u32x4r bl, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8r, ah), __builtin_bit_cast(f16x8r, bh), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8r, al), __builtin_bit_cast(f16x8r, bh), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8r, ah), __builtin_bit_cast(f16x8r, bl), c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ f32x4 mfma3_bf16(u32x4r ah, u32x4r al, u32x4r bh, u32x4r bl, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8r, ah), __builtin_bit_cast(bf16x8r, bh), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8r, al), __builtin_bit_cast(bf16x8r, bh), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8r, ah), __builtin_bit_cast(bf16x8r, bl), c, 0, 0, 0);
    return c;
}

// dL/dF of sample j, lane (j, g): channels 16 ft + 4 g .. + 3 -> the sample's slots, split into bf16 hi << 16 | bf16 lo
struct DfSlots { int2 sl[3]; };
__device__ __forceinline__ DfSlots load_df_slots(const RowsOut& ro, size_t sample) {        // early: the stores depend on it
    DfSlots d;
    const int2* ps = ro.pos + sample * ro.P;
    d.sl[0] = ps[0]; d.sl[1] = ps[1];
    d.sl[2] = ro.P > 2 ? ps[2] : make_int2(-1, -1);
    return d;
}
__device__ __forceinline__ void store_df_sorted(const RowsOut& ro, const DfSlots& ds, int g, const f32x4 dF[2]) {
    u32x4r w[2];
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
        unsigned o[4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const f32x2r x = {dF[ft][2 * q], dF[ft][2 * q + 1]};
            const unsigned hh = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2r));
            const f32x2r r = {x[0] - __builtin_bit_cast(float, hh << 16), x[1] - __builtin_bit_cast(float, hh & 0xffff0000u)};
            const unsigned ll = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2r));
            o[2 * q] = (hh << 16) | (ll & 0xffffu);
            o[2 * q + 1] = (hh & 0xffff0000u) | (ll >> 16);
        }
        w[ft] = u32x4r{o[0], o[1], o[2], o[3]};
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const int2 sl = ds.sl[pl];
        // (streaming stores: 0.8 GB per 2 frames must not push the plane bands the gather lives on out of the L2s)
        if (sl.x >= 0) {
            u32x4r* d = reinterpret_cast<u32x4r*>(ro.dfs + (size_t)sl.x * 32 + 4 * g);
            __builtin_nontemporal_store(w[0], d); __builtin_nontemporal_store(w[1], d + 4);
        }
        if (sl.y >= 0) {
            u32x4r* d = reinterpret_cast<u32x4r*>(ro.dfs + (size_t)sl.y * 32 + 4 * g);
            __builtin_nontemporal_store(w[0], d); __builtin_nontemporal_store(w[1], d + 4);
        }
    }
}

// wave-wide max of a non-negative value (all 64 lanes)
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// 2^-e with e chosen so that bound * 2^-e lies in (2^13, 2^14] (1 when the bound is 0 / not finite)
__device__ __forceinline__ float down_scale(float bound, float* up) {
    int e = 14;
    if (bound > 0.f && bound < 3.0e38f) (void)frexpf(bound, &e);      // bound = f 2^e, f in [0.5, 1)
    e = max(-60, min(60, e - 14));
    *up = ldexpf(1.f, e);
    return ldexpf(1.f, -e);
}

struct Dec16Regs {
    u32x4r w0h[4], w0l[4];            // layer 1 A operands (M tile mt): W0[16mt + j][8g + c] * sW0, hi / lo
    u32x4r w1h[2][2], w1l[2][2];      // layer 2 A operands (colour tile ot, K step ks): W1[1+16ot+j][16(2ks+(c>>2)) + 4g + (c&3)] * sW1
    float b0c[4][4], wsig[4][4], b1c[2][4];
    float bsig;
    float sF, sH;                     // down-scales of the features / hidden units
    float u1, u2;                     // up-scales of the layer-1 / layer-2 accumulators
    // the sigma row of W1 as a THIRD layer-2 M tile (row 0 = the row, rows 1..15 zero; K step ks, same K order as w1h) with its
    // own weight scale, and the up-scale of that accumulator (decoder_fwd16_l1: sigma on the matrix pipe, not a 16-FMA dot)
    u32x4r wsh[2], wsl[2];
    float u2s;
};

// from the fp32 register image + the bound M on |planes| (max over the HFAGP_ABSMAX_SLOTS slots)
__device__ __forceinline__ void make_dec16(const DecoderRegs& w, const float* planes_absmax, int lane, Dec16Regs& d) {
    const float M = wave_max(planes_absmax[lane * HFAGP_ABSMAX_STRIDE]);
    float m0 = 0.f, m1 = 0.f, rs = 0.f, hb = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float row = 0.f;                                   // sum_k |W0[16mt + j][k]| over this lane's 8 columns ...
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            m0 = fmaxf(m0, fabsf(w.w0a[mt][t]));
            row += fabsf(w.w0a[mt][t]);
        }
        row += __shfl_xor(row, 16);                        // ... and over the four column groups g
        row += __shfl_xor(row, 32);
        rs = fmaxf(rs, row);
#pragma unroll
        for (int r = 0; r < 4; ++r) hb = fmaxf(hb, fabsf(w.b0c[mt][r]));
    }
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int k = 0; k < 16; ++k) m1 = fmaxf(m1, fabsf(w.w1a[ot][k]));
    m0 = wave_max(m0); m1 = wave_max(m1); rs = wave_max(rs); hb = wave_max(hb);
    float uF, uH, uW0, uW1;
    d.sF = down_scale(M, &uF);
    d.sH = down_scale(hb + rs * M, &uH);                   // softplus(x) <= |x| + log 2 < bound + 1: the margin of 2^14 covers it
    const float sW0 = down_scale(m0, &uW0), sW1 = down_scale(m1, &uW1);
    d.u1 = uF * uW0;
    d.u2 = uH * uW1;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        float v[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = w.w0a[mt][t] * sW0;
        split8_f16(v, d.w0h[mt], d.w0l[mt]);
#pragma unroll
        for (int r = 0; r < 4; ++r) { d.b0c[mt][r] = w.b0c[mt][r]; d.wsig[mt][r] = w.wsig[mt][r]; }
    }
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = w.w1a[ot][8 * ks + c] * sW1;      // (mt = 2ks + (c>>2), r = c&3) -> index 4mt + r
            split8_f16(v, d.w1h[ot][ks], d.w1l[ot][ks]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) d.b1c[ot][r] = w.b1c[ot][r];
    }
    d.bsig = w.bsig;
    {
        float ms = 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ms = fmaxf(ms, fabsf(w.wsig[mt][r]));
        float uWs;
        const float sWs = down_scale(wave_max(ms), &uWs);
        d.u2s = uH * uWs;
        const bool row0 = (lane & 15) == 0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = row0 ? w.wsig[2 * ks + (c >> 2)][c & 3] * sWs : 0.f;
            split8_f16(v, d.wsh[ks], d.wsl[ks]);
        }
    }
}

// drop-in for decoder_fwd: same inputs / outputs / register layouts
template <bool KEEP_PRE>
__device__ __forceinline__ void decoder_fwd16(const Dec16Regs& w, const float f[8], f32x4 hp[4], f32x4 h[4],
                                              float& sigma, f32x4 o[2]) {
    u32x4r fh, fl;
    {
        float fs[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) fs[t] = f[t] * w.sF;
        split8_f16(fs, fh, fl);
    }
    float sg = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const f32x4 acc = mfma3_f16(w.w0h[mt], w.w0l[mt], fh, fl, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pre = fmaf(acc[r], w.u1, w.b0c[mt][r]);
            if (KEEP_PRE) hp[mt][r] = pre;
            h[mt][r] = softplus_f(pre);
            sg = fmaf(h[mt][r], w.wsig[mt][r], sg);
        }
    }
    sg += __shfl_xor(sg, 16);
    sg += __shfl_xor(sg, 32);
    sigma = sg + w.bsig;
    u32x4r hh[2], hl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float hs[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) hs[c] = h[2 * ks + (c >> 2)][c & 3] * w.sH;
        split8_f16(hs, hh[ks], hl[ks]);
    }
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) acc = mfma3_f16(w.w1h[ot][ks], w.w1l[ot][ks], hh[ks], hl[ks], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[ot][r] = fmaf(acc[r], w.u2, w.b1c[ot][r]);
    }
}

// Layer 1 of the 16-bit decoder from a workgroup-shared LDS image (forward kernel): the 64 registers of w0h / w0l / b0c /
// wsig become 8.5 KB of LDS — [hi|lo][mt][lane] 16-byte A operands (lane-linear: conflict-free ds_read_b128) and the compact
// bias / sigma-row tables [mt][g][r] (broadcast reads) — and pay for the 64 registers of keeping all 24 texel loads of a tile
// in flight (gather8_all).  Layer 2 stays in registers.
constexpr int kDecL1Floats = 2 * 4 * 64 * 4 + 2 * 64;            // 2176 floats = 8704 bytes
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;
__device__ __forceinline__ void store_dec16_l1(const Dec16Regs& w, float* img, int lane) {
    unsigned* u = reinterpret_cast<unsigned*>(img);
    const int g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u[((0 * 4 + mt) * 64 + lane) * 4 + q] = w.w0h[mt][q];
            u[((1 * 4 + mt) * 64 + lane) * 4 + q] = w.w0l[mt][q];
        }
        if ((lane & 15) == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                img[2048 + (mt * 4 + g) * 4 + r] = w.b0c[mt][r] * kLog2e;        // base-2 pre-activations (see prescale_dec16_l1)
                img[2048 + 64 + (mt * 4 + g) * 4 + r] = w.wsig[mt][r] * kLn2;
            }
        }
    }
}

// The forward decoder works in BASE 2: v_exp_f32 / v_log_f32 are 2^x / log2 x, so exp(x) and log(x) each cost a multiply by
// log2 e / ln 2 next to the transcendental.  With  x2 = x log2 e :
//   softplus(x) = ln 2 * [ max(x2, 0) + log2(1 + 2^-|x2|) ]        sigmoid(y) = 1 / (1 + 2^-y2)
// and the constants fold into what surrounds them: log2 e into the dequantisation scale and bias of each layer's accumulator
// (u1, b0, u2, b1), ln 2 into the consumers of the hidden activations (the sigma row and the fp16 split scale sH).
// 3 of the ~9 VALU instructions per softplus and 1 of 5 per sigmoid go away (24 values per lane per 16-sample tile).
__device__ __forceinline__ void prescale_dec16_l1(Dec16Regs& w) {     // AFTER store_dec16_l1 (which scales b0 / wsig itself)
    w.u1 *= kLog2e;
    w.u2 *= kLog2e;
    w.sH *= kLn2;
    w.sF *= 0.3333333333333333f;      // tile_reduce hands over the SUM over the three planes, not their mean
#pragma unroll
    for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int r = 0; r < 4; ++r) w.b1c[ot][r] *= kLog2e;
}
__device__ __forceinline__ float softplus2_f(float x2) {             // softplus(x) / ln 2 of x2 = x log2 e
    return fmaxf(x2, 0.f) + __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(-fabsf(x2)));
}
__device__ __forceinline__ float sigmoid2_f(float y2) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-y2)); }

// h: softplus / ln 2 of the hidden layer; o: the colour logits times log2 e (feed sigmoid2_f); sigma: on the lanes g == 0 only
__device__ __forceinline__ void decoder_fwd16_l1(const Dec16Regs& w, const float* img, int lane, const float f[8], float& sigma,
                                                 f32x4 o[2]) {
    const int g = lane >> 4;
    u32x4r fh, fl;
    {
        float fs[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) fs[t] = f[t] * w.sF;
        split8_f16(fs, fh, fl);
    }
    f32x4 h[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const u32x4r ah = *reinterpret_cast<const u32x4r*>(img + ((0 * 4 + mt) * 64 + lane) * 4);
        const u32x4r al = *reinterpret_cast<const u32x4r*>(img + ((1 * 4 + mt) * 64 + lane) * 4);
        const float4 b0 = *reinterpret_cast<const float4*>(img + 2048 + (mt * 4 + g) * 4);
        const float b0v[4] = {b0.x, b0.y, b0.z, b0.w};
        const f32x4 acc = mfma3_f16(ah, al, fh, fl, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int r = 0; r < 4; ++r) h[mt][r] = softplus2_f(fmaf(acc[r], w.u1, b0v[r]));
    }
    u32x4r hh[2], hl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float hs[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) hs[c] = h[2 * ks + (c >> 2)][c & 3] * w.sH;
        split8_f16(hs, hh[ks], hl[ks]);
    }
    {   // sigma = row 0 of a third M tile of layer 2 (6 MFMAs on a pipe that is 20 % busy instead of 16 FMAs + 2 shuffles + the
        // row's LDS reads on the vector ALU that bounds the kernel); C row 0 = register 0 of the lanes g == 0, column j = sample
        f32x4 as = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) as = mfma3_f16(w.wsh[ks], w.wsl[ks], hh[ks], hl[ks], as);
        sigma = fmaf(as[0], w.u2s, w.bsig);       // (valid on the lanes g == 0: the only ones that store it)
    }
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) acc = mfma3_f16(w.w1h[ot][ks], w.w1l[ot][ks], hh[ks], hl[ks], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[ot][r] = fmaf(acc[r], w.u2, w.b1c[ot][r]);
    }
}

// The gather with ALL 24 texel loads (3 planes x 4 taps x 2 float4) of a 16-sample tile in flight at once, in two halves so that
// the caller can put the decoder of the PREVIOUS tile between them: gather8 keeps one plane (8 loads, 32 registers) in flight at
// a time — three serial memory round trips per tile, all exposed at two waves per SIMD.
struct TileLoads {
    float4 v0[3][4], v1[3][4];
    float w[3][4];
};
__device__ __forceinline__ void tile_issue(const HfagpRaymarchArgs& a, int b, int g, const PlaneTaps taps[3], TileLoads& t) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const char* base = reinterpret_cast<const char*>(a.planes + ((size_t)(b * 3 + pl) * a.H * a.W) * 32);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned off = ((unsigned)taps[pl].idx[k] * 32u + 8u * g) * 4u;
            t.v0[pl][k] = *reinterpret_cast<const float4*>(base + off);
            t.v1[pl][k] = *reinterpret_cast<const float4*>(base + off + 16);
            t.w[pl][k] = taps[pl].w[k];
        }
    }
    __builtin_amdgcn_sched_barrier(0);          // (keep the 24 loads where they are: the scheduler sinks them to their uses otherwise)
}
// 3 x the mean over the planes (the 1/3 is folded into the feature down-scale: prescale_dec16_l1), as ONE chain of 12 FMAs per
// channel: gather8's per-plane sums + plane adds + mean cost 136 instructions per tile, this costs 96.  (A different — equally
// valid — fp32 summation order from ATen's: rounding-level differences only.)
__device__ __forceinline__ void tile_reduce(const TileLoads& t, float f[8]) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        const float4 *v0 = t.v0[pl], *v1 = t.v1[pl];
        const float v[4][8] = {{v0[0].x, v0[0].y, v0[0].z, v0[0].w, v1[0].x, v1[0].y, v1[0].z, v1[0].w},
                               {v0[1].x, v0[1].y, v0[1].z, v0[1].w, v1[1].x, v1[1].y, v1[1].z, v1[1].w},
                               {v0[2].x, v0[2].y, v0[2].z, v0[2].w, v1[2].x, v1[2].y, v1[2].z, v1[2].w},
                               {v0[3].x, v0[3].y, v0[3].z, v0[3].w, v1[3].x, v1[3].y, v1[3].z, v1[3].w}};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float acc = pl == 0 ? v[0][c] * t.w[0][0] : fmaf(v[0][c], t.w[pl][0], f[c]);
            acc = fmaf(v[1][c], t.w[pl][1], acc);
            acc = fmaf(v[2][c], t.w[pl][2], acc);
            f[c] = fmaf(v[3][c], t.w[pl][3], acc);
        }
    }
}

// ---- LDS images for the backward kernels (lane-linear rows of 64 dwords: every ds_read_b32 is conflict-free)
// Dec16Regs image: rows 0-15 w0h[mt][q], 16-31 w0l, 32-47 w1h[ot][ks][q], 48-63 w1l, 64-79 b0c[mt][r], 80-95 wsig[mt][r],
// 96-103 b1c[ot][r], 104 bsig, 105 sF, 106 sH, 107 u1, 108 u2
constexpr int kDec16LdsRows = 109, kDec16Wsig = 80;
__device__ __forceinline__ void store_dec16_lds(const Dec16Regs& w, float* imgf, int lane) {
    unsigned* img = reinterpret_cast<unsigned*>(imgf);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            img[(mt * 4 + q) * 64 + lane] = w.w0h[mt][q];
            img[(16 + mt * 4 + q) * 64 + lane] = w.w0l[mt][q];
            imgf[(64 + mt * 4 + q) * 64 + lane] = w.b0c[mt][q];
            imgf[(kDec16Wsig + mt * 4 + q) * 64 + lane] = w.wsig[mt][q];
        }
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                img[(32 + (ot * 2 + ks) * 4 + q) * 64 + lane] = w.w1h[ot][ks][q];
                img[(48 + (ot * 2 + ks) * 4 + q) * 64 + lane] = w.w1l[ot][ks][q];
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) imgf[(96 + ot * 4 + r) * 64 + lane] = w.b1c[ot][r];
    }
    imgf[104 * 64 + lane] = w.bsig;
    imgf[105 * 64 + lane] = w.sF;
    imgf[106 * 64 + lane] = w.sH;
    imgf[107 * 64 + lane] = w.u1;
    imgf[108 * 64 + lane] = w.u2;
}

__device__ __forceinline__ u32x4r lds_row4(const float* imgf, int row0, int lane) {
    const unsigned* img = reinterpret_cast<const unsigned*>(imgf) + lane;
    return u32x4r{img[row0 * 64], img[(row0 + 1) * 64], img[(row0 + 2) * 64], img[(row0 + 3) * 64]};
}

template <bool KEEP_PRE>
__device__ __forceinline__ void decoder_fwd16_lds(const float* imgf, int lane, const float f[8], f32x4 hp[4], f32x4 h[4],
                                                  float& sigma, f32x4 o[2]) {
    const float* W = imgf + lane;
    const float sF = W[105 * 64], sH = W[106 * 64], u1 = W[107 * 64], u2 = W[108 * 64];
    u32x4r fh, fl;
    {
        float fs[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) fs[t] = f[t] * sF;
        split8_f16(fs, fh, fl);
    }
    float sg = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const f32x4 acc = mfma3_f16(lds_row4(imgf, mt * 4, lane), lds_row4(imgf, 16 + mt * 4, lane), fh, fl,
                                    f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pre = fmaf(acc[r], u1, W[(64 + mt * 4 + r) * 64]);
            if (KEEP_PRE) hp[mt][r] = pre;
            h[mt][r] = softplus_f(pre);
            sg = fmaf(h[mt][r], W[(kDec16Wsig + mt * 4 + r) * 64], sg);
        }
    }
    sg += __shfl_xor(sg, 16);
    sg += __shfl_xor(sg, 32);
    sigma = sg + W[104 * 64];
    u32x4r hh[2], hl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float hs[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) hs[c] = h[2 * ks + (c >> 2)][c & 3] * sH;
        split8_f16(hs, hh[ks], hl[ks]);
    }
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            acc = mfma3_f16(lds_row4(imgf, 32 + (ot * 2 + ks) * 4, lane), lds_row4(imgf, 48 + (ot * 2 + ks) * 4, lane),
                            hh[ks], hl[ks], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[ot][r] = fmaf(acc[r], u2, W[(96 + ot * 4 + r) * 64]);
    }
}

// Gradient A operands (split bf16: a gradient needs fp32's exponent range; ~2^-16 per product is ample for a gradient):
//   g1img rows mt*4+q (hi), 16+mt*4+q (lo):        A[16mt + j][k = (g, c)] = W1[1 + 16(c>>2) + 4g + (c&3)][16mt + j] * g1
//   g0img rows (ft*2+ks)*4+q (hi), 16+... (lo):    A[16ft + j][k = (g, c)] = W0[16(2ks + (c>>2)) + 4g + (c&3)][16ft + j] * g0
// built by whichever waves the block has (`wave` of `nwaves`)
__device__ __forceinline__ void build_grad16_lds(const HfagpRaymarchArgs& a, float* g1img, float* g0img, int lane, int wave,
                                                 int nwaves) {
    const int j = lane & 15, g = lane >> 4;
    const float g0 = a.decoder_lr_mul * 0.17677669529663687f, g1 = a.decoder_lr_mul * 0.125f;
    unsigned* i1 = reinterpret_cast<unsigned*>(g1img);
    unsigned* i0 = reinterpret_cast<unsigned*>(g0img);
    for (int t = wave; t < 8; t += nwaves) {
        float v[8];
        u32x4r hi, lo;
        if (t < 4) {
            const int mt = t;
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = a.dec_w1[(1 + 16 * (c >> 2) + 4 * g + (c & 3)) * 64 + 16 * mt + j] * g1;
            split8_bf16(v, hi, lo);
#pragma unroll
            for (int q = 0; q < 4; ++q) { i1[(mt * 4 + q) * 64 + lane] = hi[q]; i1[(16 + mt * 4 + q) * 64 + lane] = lo[q]; }
        } else {
            const int ft = (t - 4) >> 1, ks = (t - 4) & 1;
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = a.dec_w0[(16 * (2 * ks + (c >> 2)) + 4 * g + (c & 3)) * 32 + 16 * ft + j] * g0;
            split8_bf16(v, hi, lo);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                i0[((ft * 2 + ks) * 4 + q) * 64 + lane] = hi[q];
                i0[(16 + (ft * 2 + ks) * 4 + q) * 64 + lane] = lo[q];
            }
        }
    }
}

// decoder backward on the 16-bit pipe: dH (pre-activation gradient, C layout of the hidden tiles) and dF (C layout of the
// two feature tiles) from dO (colour-logit gradients), d sigma, the saved pre-activations
__device__ __forceinline__ void decoder_bwd16_lds(const float* dimg, const float* g1img, const float* g0img, int lane,
                                                  const f32x4 dO[2], float dsig, const f32x4 hp[4], f32x4 dH[4], f32x4 dF[2]) {
    u32x4r oh, ol;
    {
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = dO[c >> 2][c & 3];
        split8_bf16(v, oh, ol);
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const float* ws_ = dimg + (kDec16Wsig + mt * 4) * 64 + lane;
        dH[mt] = f32x4{ws_[0] * dsig, ws_[64] * dsig, ws_[128] * dsig, ws_[192] * dsig};
        dH[mt] = mfma3_bf16(lds_row4(g1img, mt * 4, lane), lds_row4(g1img, 16 + mt * 4, lane), oh, ol, dH[mt]);
#pragma unroll
        for (int r = 0; r < 4; ++r) dH[mt][r] *= sigmoid_f(hp[mt][r]);      // softplus' = sigmoid
    }
    u32x4r dh[2], dl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = dH[2 * ks + (c >> 2)][c & 3];
        split8_bf16(v, dh[ks], dl[ks]);
    }
#pragma unroll
    for (int ft = 0; ft < 2; ++ft) {
        dF[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            dF[ft] = mfma3_bf16(lds_row4(g0img, (ft * 2 + ks) * 4, lane), lds_row4(g0img, 16 + (ft * 2 + ks) * 4, lane),
                                dh[ks], dl[ks], dF[ft]);
    }
}

// The same decoder with the A operands read from an LDS image [105][64] of DecoderRegs (lane-linear, so every
// ds_read_b32 is conflict-free): used where the registers are needed for the backward products.
constexpr int kDecLdsRows = 109;          // (109: the 16-bit image, kDec16LdsRows, shares the buffer)
__device__ __forceinline__ void store_decoder_lds(const DecoderRegs& w, float* img, int lane) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int t = 0; t < 8; ++t) img[(mt * 8 + t) * 64 + lane] = w.w0a[mt][t];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            img[(32 + mt * 4 + r) * 64 + lane] = w.b0c[mt][r];
            img[(48 + mt * 4 + r) * 64 + lane] = w.wsig[mt][r];
        }
    }
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
#pragma unroll
        for (int k = 0; k < 16; ++k) img[(64 + ot * 16 + k) * 64 + lane] = w.w1a[ot][k];
#pragma unroll
        for (int r = 0; r < 4; ++r) img[(96 + ot * 4 + r) * 64 + lane] = w.b1c[ot][r];
    }
    img[104 * 64 + lane] = w.bsig;
}

template <bool KEEP_PRE>
__device__ __forceinline__ void decoder_fwd_lds(const float* img, int lane, const float f[8], f32x4 hp[4], f32x4 h[4],
                                                float& sigma, f32x4 o[2]) {
    const float* W = img + lane;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        h[mt] = f32x4{W[(32 + mt * 4) * 64], W[(33 + mt * 4) * 64], W[(34 + mt * 4) * 64], W[(35 + mt * 4) * 64]};
#pragma unroll
        for (int t = 0; t < 8; ++t)
            h[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[(mt * 8 + t) * 64], f[t], h[mt], 0, 0, 0);
    }
    float sg = 0.f;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (KEEP_PRE) hp[mt][r] = h[mt][r];
            h[mt][r] = softplus_f(h[mt][r]);
            sg = fmaf(h[mt][r], W[(48 + mt * 4 + r) * 64], sg);
        }
    sg += __shfl_xor(sg, 16);
    sg += __shfl_xor(sg, 32);
    sigma = sg + W[104 * 64];
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
        o[ot] = f32x4{W[(96 + ot * 4) * 64], W[(97 + ot * 4) * 64], W[(98 + ot * 4) * 64], W[(99 + ot * 4) * 64]};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                o[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[(64 + ot * 16 + mt * 4 + r) * 64], h[mt][r], o[ot], 0, 0, 0);
    }
}

}  // namespace hfagp

// Ray-march backward, pass 2 as SORT + GATHER (gfx950): d planes without the per-sample scatter.
//
// The scatter forms (raymarch_bwd.hip) push every sample's dL/dF line into 8-12 texel lines of d_planes: ~10 M fp32 line
// atomics per 2 frames (fabric transactions: 1.1 GB of write traffic for a 50 MB output) or a walk through an LDS line cache.
// Here the samples are SORTED by where they land and every output row is produced by dense matrix products:
//
//   bin    = (frame, plane, 32-texel column strip xb, texel row y0 = floor(iy))        P * ceil(W/32) * (H+1) bins per frame
//            a sample of bin (xb, y0) touches rows y0 (weight 1 - wy) and y0 + 1 (weight wy) of the strip, at the columns
//            floor(ix), floor(ix) + 1 (a sample whose column pair straddles two strips is entered in both);
//   S1     raymarch_bwd_bins_kernel<false>: per 32-ray column chunk an LDS histogram of the bins, added to the global counts;
//   scan   raymarch_bwd_scan_kernel: bin -> slot offset (bins padded to 16 slots), and the list of units = bin chunks of
//          <= 1024 slots;
//   S2     raymarch_bwd_bins_kernel<true>: reserves the chunk's range of every bin with ONE global atomic per bin, ranks its
//          samples inside with LDS atomics, writes (ix, wy) to the slot and the slot number to pos[sample][plane];
//   dF     raymarch_bwd_df_kernel / raymarch_bwd_tiles_kernel<PG> (raymarch_bwd.hip) write dL/dF of a sample, split into
//          bf16 hi | lo, straight to its slots: the gather below then STREAMS contiguous memory;
//   G      raymarch_bwd_rows_kernel: one wave per unit;  per batch of 16 slots
//              A[m][k] = hat(ix_k - x_m) * rowweight_k / 3     32 texels x 16 samples, built in registers, split bf16 hi + lo
//              B[k][c] = dL/dF[k][c]                            16 samples x 32 channels, v_perm of the packed words
//              acc_row(y0), acc_row(y0+1) += A . B              2 x 3 v_mfma_f32_32x32x16_bf16
//          hat(t) = max(0, 1 - |t|) IS the bilinear weight of grid_sample (bit for bit: ix - x0 and ix - (x0 + 1) are exact),
//          and a texel outside the plane simply is no row / column of any strip (zeros padding).
//          At the end the two 32 x 32 row tiles are added to d_planes: two addends per output element and bin chunk
//          (~0.7 M line atomics per 2 frames instead of ~10 M), nothing per sample.
#include <algorithm>
#include <cstdlib>
#include "raymarch_common.h"

namespace hfagp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kRowsStrip = 32;          // texel columns per strip = M of the MFMA
constexpr int kRowsUnit = 1024;         // slots per unit (a multiple of 16)
constexpr int kRowsMaxBins = 8192;      // bins per frame the two LDS tables of S2 hold (64 KB)
constexpr int kRowsRays = 32;           // rays of a column chunk (S1 / S2 block)

struct RowsDev {
    unsigned* cnt;       // [NB]      entries per bin
    unsigned* off;       // [NB + 1]  first slot of the bin (bins padded to 16 slots)
    unsigned* cursor;    // [NB]      S2: next free slot of the bin
    unsigned* meta;      // [4]       number of units, number of slots
    int4* units;         // [max_units] (bin, first slot, end slot (a multiple of 16 slots from the first), end of the bin's entries)
    int2* pos;           // [samples][P] slot of the sample in plane p (and of its second entry), -1 = none
    float2* ent;         // [cap] (column coordinate ix, row fraction wy) of the slot's sample
    unsigned* dfs;       // [cap][32] dL/dF, bf16 hi << 16 | bf16 lo
    int P, XB, H1, NBF, NB, max_units, unit;
    unsigned cap;
};

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// carve the scratch buffer; returns the bytes needed (base may be null), 0 when the variant does not apply
static size_t rows_layout(const HfagpRaymarchArgs& a, unsigned char* base, RowsDev& d) {
    const bool mirror = a.plane_axes == 0 && a.H == a.W;
    d.P = mirror ? 2 : 3;
    d.XB = (a.W + kRowsStrip - 1) / kRowsStrip;
    d.H1 = a.H + 1;
    const long long nbf = (long long)d.P * d.XB * d.H1;
    if (nbf > kRowsMaxBins) return 0;
    d.NBF = (int)nbf;
    const long long nb = nbf * a.B;
    const long long samples = (long long)a.B * a.res * a.res * (a.Sc + a.Sf);
    const long long cap = 2ll * d.P * samples + 16ll * nb;
    if (nb > (1ll << 24) || cap >= (1ll << 31)) return 0;
    d.NB = (int)nb;
    d.cap = (unsigned)cap;
    static const int unit_env = getenv("HFAGP_DEV_ROWS_UNIT") ? atoi(getenv("HFAGP_DEV_ROWS_UNIT")) : 0;      // developer: A/B timing
    d.unit = unit_env >= 16 ? unit_env & ~15 : kRowsUnit;
    d.max_units = (int)(cap / d.unit + nb);
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o = align256(o + bytes); return base ? base + at : nullptr; };
    d.cnt = reinterpret_cast<unsigned*>(take((size_t)nb * 4));
    d.meta = reinterpret_cast<unsigned*>(take(16));
    d.off = reinterpret_cast<unsigned*>(take((size_t)(nb + 1) * 4));
    d.cursor = reinterpret_cast<unsigned*>(take((size_t)nb * 4));
    d.units = reinterpret_cast<int4*>(take((size_t)d.max_units * 16));
    d.pos = reinterpret_cast<int2*>(take((size_t)samples * d.P * 8));
    d.ent = reinterpret_cast<float2*>(take((size_t)cap * 8));
    d.dfs = reinterpret_cast<unsigned*>(take((size_t)cap * 128));
    return o;
}

struct RowsKey { int key, dup; float ix, wy; };       // bin of the frame (-1: the sample has no texel in this plane)

__device__ __forceinline__ RowsKey rows_key(const HfagpRaymarchArgs& a, const RowsDev& r, const float q[3], int pl) {
    float gx, gy, ix, iy;
    plane_coords(a, q, pl, gx, gy);
    plane_pixel(a, gx, gy, ix, iy);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fminf(fmaxf(fx0, -2.f), (float)a.W + 1.f), y0 = (int)fminf(fmaxf(fy0, -2.f), (float)a.H + 1.f);
    RowsKey k;
    k.key = -1; k.dup = -1; k.ix = ix; k.wy = __fsub_rn(iy, fy0);
    if (x0 >= -1 && x0 < a.W && y0 >= -1 && y0 < a.H) {
        const int xb = max(x0, 0) >> 5;
        k.key = (pl * r.XB + xb) * r.H1 + y0 + 1;
        if (x0 >= 0 && (x0 & 31) == 31 && x0 + 1 < a.W) k.dup = k.key + r.H1;       // columns x0 | x0 + 1 in two strips
    }
    return k;
}

// S1 (PLACE = false) / S2 (PLACE = true): a workgroup = 32 rays of one image column (their samples fall into few bins:
// one or two strips of every plane), thread = (ray, every 8th sample)
template <int S, bool PLACE>
__global__ void __launch_bounds__(256)
raymarch_bwd_bins_kernel(const RayParams p, const RowsDev r, const int chunks_per_col) {
    extern __shared__ unsigned rows_smem[];
    unsigned* hist = rows_smem;                    // [NBF]
    unsigned* base = rows_smem + r.NBF;            // [NBF], PLACE only
    const HfagpRaymarchArgs& a = p.a;
    const int chunk = blockIdx.x;
    const int col_id = chunk / chunks_per_col, piece = chunk % chunks_per_col;
    const int b = col_id / a.res, pj = col_id % a.res;
    const int pi = piece * kRowsRays + (threadIdx.x >> 3), sub = threadIdx.x & 7;
    const bool active = pi < a.res;
    for (int i = threadIdx.x; i < r.NBF; i += 256) hist[i] = 0;
    __syncthreads();
    float o3[3], d3[3];
    ray_setup(a, b, min(pi, a.res - 1), pj, o3, d3);
    const size_t ray = (size_t)b * a.res * a.res + (size_t)min(pi, a.res - 1) * a.res + pj;
    // the thread's S / 8 depths: all loads go out together and stay in registers for the second pass of S2
    const float* depth = p.rec + (ray * S + sub) * 4;
    float dep[S / 8];
#pragma unroll
    for (int i = 0; i < S / 8; ++i) dep[i] = depth[i * 32];
    if (active) {
#pragma unroll 2
        for (int i = 0; i < S / 8; ++i) {
            float q[3];
            sample_point(p, o3, d3, dep[i], q);
            for (int pl = 0; pl < r.P; ++pl) {
                const RowsKey k = rows_key(a, r, q, pl);
                if (k.key >= 0) atomicAdd(&hist[k.key], 1u);
                if (k.dup >= 0) atomicAdd(&hist[k.dup], 1u);
            }
        }
    }
    __syncthreads();
    if constexpr (!PLACE) {
        for (int i = threadIdx.x; i < r.NBF; i += 256) {
            const unsigned c = hist[i];
            if (c) atomicAdd(&r.cnt[(size_t)b * r.NBF + i], c);
        }
    } else {
        // one RETURNING global atomic per bin the chunk touches (~600 of them): four in flight per thread — one at a time the
        // loop waited out 16 atomic round trips in sequence (S2 76 us, half of it here)
        for (int i0 = threadIdx.x; i0 < r.NBF; i0 += 256 * 4) {
            unsigned c[4], got[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] = i0 + 256 * k < r.NBF ? hist[i0 + 256 * k] : 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) got[k] = c[k] ? atomicAdd(&r.cursor[(size_t)b * r.NBF + i0 + 256 * k], c[k]) : 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i0 + 256 * k < r.NBF) { base[i0 + 256 * k] = got[k]; hist[i0 + 256 * k] = 0; }
        }
        __syncthreads();
        if (active) {
#pragma unroll 2
            for (int i = 0; i < S / 8; ++i) {
                const int s = sub + 8 * i;
                float q[3];
                sample_point(p, o3, d3, dep[i], q);
                for (int pl = 0; pl < r.P; ++pl) {
                    const RowsKey k = rows_key(a, r, q, pl);
                    int2 sl = make_int2(-1, -1);
                    if (k.key >= 0) {
                        sl.x = (int)(base[k.key] + atomicAdd(&hist[k.key], 1u));
                        r.ent[sl.x] = make_float2(k.ix, k.wy);
                    }
                    if (k.dup >= 0) {
                        sl.y = (int)(base[k.dup] + atomicAdd(&hist[k.dup], 1u));
                        r.ent[sl.y] = make_float2(k.ix, k.wy);
                    }
                    r.pos[(ray * S + s) * r.P + pl] = sl;
                }
            }
        }
    }
}

// counts -> slot offsets (bins padded to 16 slots) + the unit list; one workgroup.  Tiles of 12288 bins go through LDS: coalesced
// loads / stores, the per-thread segments (12 consecutive bins) are walked in LDS.
constexpr int kScanTile = 12288;      // 12 bins per thread: 2 frames of mirrored 256^2 planes (8224 bins) are one tile
__global__ void __launch_bounds__(1024) raymarch_bwd_scan_kernel(const RowsDev r) {
    __shared__ unsigned sc[kScanTile], ssl[16], ssu[16];
    const int t = threadIdx.x;
    unsigned carry_l = 0, carry_u = 0;
    for (int t0 = 0; t0 < r.NB; t0 += kScanTile) {
        const int nt = min(kScanTile, r.NB - t0);
#pragma unroll
        for (int i = 0; i < kScanTile / 1024; ++i) sc[t + 1024 * i] = t + 1024 * i < nt ? r.cnt[t0 + t + 1024 * i] : 0u;
        __syncthreads();
        constexpr int PT = kScanTile / 1024;
        unsigned c[PT], sl = 0, su = 0;
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            c[i] = sc[PT * t + i];
            const unsigned c16 = (c[i] + 15u) & ~15u;
            sl += c16;
            su += (c16 + r.unit - 1) / r.unit;
        }
        // block scan: inside the wave with shuffles, then over the 16 wave totals (a 10-step scan in LDS costs 20 barriers of
        // 16 waves: 12 of the kernel's 17 us)
        unsigned il = sl, iu = su;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned vl = __shfl_up(il, o), vu = __shfl_up(iu, o);
            if ((t & 63) >= o) { il += vl; iu += vu; }
        }
        if ((t & 63) == 63) { ssl[t >> 6] = il; ssu[t >> 6] = iu; }
        __syncthreads();
        unsigned wl = 0, wu = 0, tl = 0, tu = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            if (w < (t >> 6)) { wl += ssl[w]; wu += ssu[w]; }
            tl += ssl[w]; tu += ssu[w];
        }
        unsigned ol = carry_l + wl + il - sl, ou = carry_u + wu + iu - su;
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const unsigned c16 = (c[i] + 15u) & ~15u;
            sc[PT * t + i] = ol;
            for (unsigned k = 0; k * r.unit < c16; ++k)
                r.units[ou++] = make_int4(t0 + PT * t + i, (int)(ol + k * r.unit), (int)(ol + min(c16, (k + 1) * r.unit)), (int)(ol + c[i]));
            ol += c16;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kScanTile / 1024; ++i)
            if (t + 1024 * i < nt) { r.off[t0 + t + 1024 * i] = sc[t + 1024 * i]; r.cursor[t0 + t + 1024 * i] = sc[t + 1024 * i]; }
        carry_l += tl; carry_u += tu;
        __syncthreads();
    }
    if (t == 0) { r.off[r.NB] = carry_l; r.meta[0] = carry_u; r.meta[1] = carry_l; }
}

// G: one wave per unit.  The stream of a unit is contiguous: per batch of 16 slots 128 bytes of (ix, wy) + 2 KB of dL/dF; a
// wave keeps kRowsDepth batches in flight (9 registers each: the 16 (ix, wy) pairs travel as ONE dword per lane and are
// re-distributed through 128 bytes of LDS when the batch is consumed) — with one batch in flight per wave the kernel ran at the
// memory LATENCY (3.9 us per batch and wave, 3.4 TB/s).
constexpr int kRowsDepth = 4;

__global__ void __launch_bounds__(256, 2)
raymarch_bwd_rows_kernel(const RowsDev r, float* __restrict__ d_planes, const int H, const int W) {
    __shared__ __attribute__((aligned(16))) float stage_all[4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned unit = blockIdx.x * 4 + wave;
    if (unit >= r.meta[0]) return;
    float* stage = stage_all[wave];
    const int4 u = r.units[unit];
    const int bin = __builtin_amdgcn_readfirstlane(u.x);
    const unsigned e0 = __builtin_amdgcn_readfirstlane(u.y), e1 = __builtin_amdgcn_readfirstlane(u.z);
    const unsigned ev = __builtin_amdgcn_readfirstlane(u.w);             // slots >= ev are the bin's padding: never written
    const int y0 = bin % r.H1 - 1, strip = bin / r.H1, xb = strip % r.XB, bp = strip / r.XB;
    const int n = lane & 31, h = lane >> 5;
    const float xm = (float)(xb * kRowsStrip + n);
    // lane (n, h): A row m = n (texel column), B column n (channel), K = samples 8h .. 8h + 7 of the batch
    const float* entp = reinterpret_cast<const float*>(r.ent) + n;
    const unsigned* dfp = r.dfs + (size_t)(8 * h) * 32 + n;
    struct Batch { float m; unsigned pk[8]; };
    auto fetch = [&](unsigned e) __attribute__((always_inline)) {
        Batch b;
        b.m = __builtin_nontemporal_load(entp + (size_t)e * 2);
        const unsigned* d = dfp + (size_t)e * 32;
#pragma unroll
        for (int t = 0; t < 8; ++t) b.pk[t] = __builtin_nontemporal_load(d + t * 32);
        return b;
    };
    f32x16 acc_u, acc_l;
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc_u[i] = 0.f; acc_l[i] = 0.f; }
    auto consume = [&](Batch cur, unsigned ec) __attribute__((always_inline)) {
        stage[n] = cur.m;
        WAVE_SYNC();
        float4 pr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pr[i] = *reinterpret_cast<const float4*>(stage + 16 * h + 4 * i);
        WAVE_SYNC();
        const float ixs[8] = {pr[0].x, pr[0].z, pr[1].x, pr[1].z, pr[2].x, pr[2].z, pr[3].x, pr[3].z};
        const float wys[8] = {pr[0].y, pr[0].w, pr[1].y, pr[1].w, pr[2].y, pr[2].w, pr[3].y, pr[3].w};
        float au[8], al[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float hx = fmaxf(1.f - fabsf(ixs[t] - xm), 0.f);
            au[t] = (1.f - wys[t]) * hx * 0.3333333333333333f;
            al[t] = wys[t] * hx * 0.3333333333333333f;
        }
        if (ec + 16 > ev) {                          // the bin's last batch: its padding slots hold whatever the buffer held
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (ec + 8 * h + t >= ev) { au[t] = 0.f; al[t] = 0.f; cur.pk[t] = 0u; }
        }
        u32x4r auh, aul, alh, all_, bh, bl;
        split8_bf16(au, auh, aul);
        split8_bf16(al, alh, all_);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bh[q] = __builtin_amdgcn_perm(cur.pk[2 * q + 1], cur.pk[2 * q], 0x07060302u);
            bl[q] = __builtin_amdgcn_perm(cur.pk[2 * q + 1], cur.pk[2 * q], 0x05040100u);
        }
        const bf16x8r vbh = __builtin_bit_cast(bf16x8r, bh), vbl = __builtin_bit_cast(bf16x8r, bl);
        acc_u = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8r, auh), vbh, acc_u, 0, 0, 0);
        acc_l = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8r, alh), vbh, acc_l, 0, 0, 0);
        acc_u = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8r, aul), vbh, acc_u, 0, 0, 0);
        acc_l = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8r, all_), vbh, acc_l, 0, 0, 0);
        acc_u = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8r, auh), vbl, acc_u, 0, 0, 0);
        acc_l = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8r, alh), vbl, acc_l, 0, 0, 0);
    };
    Batch q[kRowsDepth];
#pragma unroll
    for (int i = 0; i < kRowsDepth; ++i) q[i] = fetch(min(e0 + 16u * i, e1 - 16));
    for (unsigned e = e0; e < e1; e += 16 * kRowsDepth) {
#pragma unroll
        for (int i = 0; i < kRowsDepth; ++i) {
            const unsigned ec = e + 16 * i;
            if (ec < e1) consume(q[i], ec);                                    // (wave-uniform; no load inside the branch)
            q[i] = fetch(min(ec + 16 * kRowsDepth, e1 - 16));             // past the end: the last batch once more
        }
    }
    // C layout: lane (n, h), register i -> texel column 8 (i / 4) + 4 h + i % 4, channel n: two full 128-byte lines per store
    const int b = bp / r.P, pl = bp % r.P;
    float* plane = d_planes + ((size_t)(b * 3 + pl) * H * W) * 32 + n;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int row = y0 + half;
        if (row < 0 || row >= H) continue;
        float* dst = plane + ((size_t)row * W + xb * kRowsStrip) * 32;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = 8 * (i >> 2) + 4 * h + (i & 3);
            if (xb * kRowsStrip + m < W) unsafeAtomicAdd(dst + m * 32, half ? acc_l[i] : acc_u[i]);
        }
    }
}

size_t rows_scratch_bytes(const HfagpRaymarchArgs& a) {
    RowsDev d;
    return rows_layout(a, nullptr, d);
}

template <int S>
static void launch_bins(const RayParams& p, const RowsDev& d, int nchunks, int chunks_per_col, hipStream_t s) {
    raymarch_bwd_bins_kernel<S, false><<<nchunks, 256, (size_t)d.NBF * 4, s>>>(p, d, chunks_per_col);
    raymarch_bwd_scan_kernel<<<1, 1024, 0, s>>>(d);
    raymarch_bwd_bins_kernel<S, true><<<nchunks, 256, (size_t)d.NBF * 8, s>>>(p, d, chunks_per_col);
}

int rows_prepare(const RayParams& p, void* scratch, size_t bytes, RowsOut& out, hipStream_t s) {
    RowsDev d;
    const size_t need = rows_layout(p.a, reinterpret_cast<unsigned char*>(scratch), d);
    HFAGP_REQUIRE(need != 0 && bytes >= need, HFAGP_EBADARG, "raymarch_bwd: rows_scratch of %zu bytes, %zu needed", bytes, need);
    static bool raised = false;
    if (!raised) {
        const int lds = kRowsMaxBins * 8;
        hipError_t e = hipSuccess;
        const void* fns[] = {reinterpret_cast<const void*>(&raymarch_bwd_bins_kernel<96, true>),
                             reinterpret_cast<const void*>(&raymarch_bwd_bins_kernel<64, true>),
                             reinterpret_cast<const void*>(&raymarch_bwd_bins_kernel<32, true>)};
        for (const void* f : fns)
            if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            set_error("raymarch_bwd: cannot raise dynamic LDS to %d bytes: %s", lds, hipGetErrorString(e));
            return HFAGP_ELAUNCH;
        }
        raised = true;
    }
    hipError_t e = hipMemsetAsync(d.cnt, 0, (size_t)d.NB * 4, s);
    if (e != hipSuccess) {
        set_error("raymarch_bwd: hipMemsetAsync: %s", hipGetErrorString(e));
        return HFAGP_ELAUNCH;
    }
    const int S = p.a.Sc + p.a.Sf;
    const int chunks_per_col = (p.a.res + kRowsRays - 1) / kRowsRays;
    const int nchunks = p.a.B * p.a.res * chunks_per_col;
    if (S == 96) launch_bins<96>(p, d, nchunks, chunks_per_col, s);
    else if (S == 64) launch_bins<64>(p, d, nchunks, chunks_per_col, s);
    else launch_bins<32>(p, d, nchunks, chunks_per_col, s);
    out.pos = d.pos;
    out.dfs = d.dfs;
    out.P = d.P;
    return check_launch("raymarch_bwd/bins");
}

int rows_gather(const RayParams& p, void* scratch, float* d_planes, hipStream_t s) {
    RowsDev d;
    rows_layout(p.a, reinterpret_cast<unsigned char*>(scratch), d);
    raymarch_bwd_rows_kernel<<<(unsigned)((d.max_units + 3) / 4), 256, 0, s>>>(d, d_planes, p.a.H, p.a.W);
    return check_launch("raymarch_bwd/rows");
}

}  // namespace hfagp

using namespace hfagp;

extern "C" size_t hfagp_raymarch_bwd_rows_bytes(const HfagpRaymarchArgs* fwd) {
    if (!fwd || fwd->B <= 0 || fwd->H <= 1 || fwd->W <= 1 || fwd->res <= 0) return 0;
    return rows_scratch_bytes(*fwd);
}

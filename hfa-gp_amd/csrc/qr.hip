// Householder QR of the tall-skinny latent basis, A = (bases + 1e-8)^T  [7168 x K<=64], without touching A's
// 7168 rows in the serial part (gfx950).  SURVEY.md section 8f-4; replaces the ~250 rocSOLVER launches of
// torch.linalg.qr in HeadNeRF_*.get_latent (/root/reference/code/networks/headnerf.py:85-91).
//
// LAPACK's geqrf applies K reflections H_j = I - tau_j v_j v_j^T.  Everything step j needs is
//   * the pivot            p      = A_j[j][j]                          (A_j = H_{j-1} ... H_1 A)
//   * the column norm      |x|^2  = G_j[j][j],   G_j = Gram matrix of rows >= j of A_j
//   * w = A_j[j:, :]^T v_j        = G_j[:, j] - beta A_j[j, :]^T,      beta = -sign(p) |x|   (= R[j][j])
// and then only the top K x K block T of A_j and G_j have to be updated:
//   T[i, :] -= tau v_j[i] w^T  (i >= j),      G_{j+1} = G_j - r r^T,  r = row j of the updated T (= R[j, :]),
// because the Gram matrix of rows >= j is invariant under H_j.  So the host forms G = A^T A (one GEMM), this
// kernel runs the K steps on two K x K matrices in LDS and returns R (LAPACK's signs) and R^-1, and the host
// forms Q = A R^-1 (one GEMM).  Cost ~0.1 ms instead of ~1.4 ms; in fp32 the result is closer to the fp64
// factorisation than rocSOLVER's (the basis is well conditioned; accuracy degrades like cond(A)^2 * eps).
#include "common.h"

namespace hfagp {

constexpr int kQrMax = 64;
constexpr int kQrPerThread = kQrMax * kQrMax / 256;          // elements of a K x K matrix per thread of the 256

__global__ void __launch_bounds__(256) qr_gram_kernel(const float* __restrict__ G_in, const float* __restrict__ T_in,
                                                      float* __restrict__ R_out, float* __restrict__ Rinv_out, int n) {
    __shared__ float G[kQrMax][kQrMax + 1], T[kQrMax][kQrMax + 1], Ri[kQrMax][kQrMax + 1];
    __shared__ float w[kQrMax], v[kQrMax], r[kQrMax];
    const int tid = threadIdx.x;
    for (int i = tid; i < n * n; i += 256) {
        G[i / n][i % n] = G_in[i];
        T[i / n][i % n] = T_in[i];
    }
    int ea[kQrPerThread], ec[kQrPerThread];                    // this thread's elements e = tid + 256 k -> (row, column)
#pragma unroll
    for (int k = 0; k < kQrPerThread; ++k) {
        const int e = tid + 256 * k;
        ea[k] = e < n * n ? e / n : n;                          // (row n: out of range, skipped)
        ec[k] = e < n * n ? e % n : 0;
    }
    __syncthreads();
    for (int j = 0; j < n; ++j) {
        // scalars of the step (every thread reads the same LDS words)
        const float p = T[j][j], norm2 = G[j][j];
        const float xnorm2 = norm2 - p * p;                     // |x[1:]|^2
        const bool reflect = xnorm2 > 0.f && norm2 > 0.f;       // LAPACK larfg: xnorm == 0 -> H = I
        const float beta = reflect ? (p >= 0.f ? -sqrtf(norm2) : sqrtf(norm2)) : p;
        const float tau = reflect ? 1.f / (norm2 - beta * p) : 0.f;       // 2 / (v^T v), v = x - beta e_j
        // round 4: TWO barriers per step instead of three — the new row j of T (= row j of R, the vector the Gram matrix is
        // down-dated with) is formed here from the OLD values together with w and v, so the update of T's rows > j and the
        // down-date of G no longer wait for each other
        if (tid < n) {
            const float wt = G[tid][j] - beta * T[j][tid];
            w[tid] = wt;
            v[tid] = tid == j ? p - beta : (tid > j ? T[tid][j] : 0.f);
            r[tid] = tid == j ? beta : T[j][tid] - tau * (p - beta) * wt;     // (exact diagonal: the update would round it)
        }
        __syncthreads();
        // one pass over the thread's FIXED elements (row / column computed once, outside the step loop: the two integer
        // divisions per element were most of a step — 3.3 us per step, 165 us per call)
        // (all reads first, then all writes: a thread's elements are its own, but the compiler cannot know that T[a][c] of one
        // iteration is not T[a'][c'] of the next and would serialise 32 LDS round trips per step)
        float tv[kQrPerThread], gv[kQrPerThread];
#pragma unroll
        for (int k = 0; k < kQrPerThread; ++k) {
            const int a = min(ea[k], n - 1), c = ec[k];
            tv[k] = T[a][c];
            gv[k] = G[a][c];
        }
#pragma unroll
        for (int k = 0; k < kQrPerThread; ++k) {
            const int a = ea[k], c = ec[k];
            if (a >= j && a < n) {
                T[a][c] = a == j ? r[c] : tv[k] - tau * v[a] * w[c];
                if (c >= j) G[a][c] = gv[k] - r[a] * r[c];
            }
        }
        __syncthreads();
    }
    // R = triu(T);  R^-1 by back substitution, one column per thread
    for (int e = tid; e < n * n; e += 256) {
        const int i = e / n, c = e % n;
        R_out[e] = c >= i ? T[i][c] : 0.f;
    }
    if (tid < n) {
        const int c = tid;
        for (int i = n - 1; i >= 0; --i) {
            // (eight independent partial sums: the loads of a group are issued together — one LDS round trip per 8 terms
            // instead of one per term; this serial tail was ~50 us of the kernel)
            float pa[8] = {i == c ? 1.f : 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int k = i + 1;
            for (; k + 8 <= c + 1; k += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) pa[u] -= T[i][k + u] * Ri[k + u][c];
            }
            for (; k <= c; ++k) pa[0] -= T[i][k] * Ri[k][c];
            const float acc = ((pa[0] + pa[1]) + (pa[2] + pa[3])) + ((pa[4] + pa[5]) + (pa[6] + pa[7]));
            Ri[i][c] = i <= c ? acc / T[i][i] : 0.f;
        }
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += 256) Rinv_out[e] = Ri[e / n][e % n];
}

// Re-orthogonalisation pass (CholeskyQR2's second step): the Gram-matrix method above leaves an orthogonality defect
// |Q1^T Q1 - I| ~ cond(A)^2 eps (2.7e-3 at cond 476, 3.2e-2 at cond 1560 in fp32).  G2 = Q1^T Q1 is then within that
// defect of I, so its Cholesky factor R2 (positive diagonal: the column signs LAPACK chose in pass 1 are kept) is
// perfectly conditioned, and Q = Q1 R2^-1 is orthonormal to O(eps).  One block: upper Cholesky G2 = R2^T R2 and
// R2^-1 in LDS.  status[0] = max |G2 - I| (the defect of pass 1: the host's conditioning monitor), status[1] = 1 if a
// pivot was not positive (the factorisation broke down: cond(A)^2 eps >~ 1; the outputs are then NaN, never silently
// wrong).
__global__ void __launch_bounds__(256) qr_refine_kernel(const float* __restrict__ G_in, float* __restrict__ R_out,
                                                        float* __restrict__ Rinv_out, float* __restrict__ status, int n) {
    __shared__ float G[kQrMax][kQrMax + 1], Ri[kQrMax][kQrMax + 1];
    __shared__ float red[256];
    __shared__ int bad;
    const int tid = threadIdx.x;
    float defect = 0.f;
    if (tid == 0) bad = 0;
    for (int i = tid; i < n * n; i += 256) {
        const float g = G_in[i];
        G[i / n][i % n] = g;
        const float d = fabsf(g - (i / n == i % n ? 1.f : 0.f));
        defect = (d > defect || d != d) ? d : defect;          // NaN propagates
    }
    red[tid] = defect;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const float o = red[tid + s];
            if (o > red[tid] || o != o) red[tid] = o;
        }
        __syncthreads();
    }
    int ea[kQrPerThread], ec[kQrPerThread];
#pragma unroll
    for (int k = 0; k < kQrPerThread; ++k) {
        const int e = tid + 256 * k;
        ea[k] = e < n * n ? e / n : n;
        ec[k] = e < n * n ? e % n : 0;
    }
    // right-looking upper Cholesky: row j of R = G[j][j:] / sqrt(G[j][j]); trailing block -= r r^T
    for (int j = 0; j < n; ++j) {
        const float piv = G[j][j];
        if (!(piv > 0.f)) { if (tid == 0) bad = 1; }
        const float inv = 1.f / sqrtf(piv);
        __syncthreads();
        if (tid >= j && tid < n) G[j][tid] *= inv;              // (G[j][j] becomes sqrt(piv))
        __syncthreads();
        float gv[kQrPerThread];                                // (fixed elements per thread; reads first, then writes: see qr_gram_kernel)
#pragma unroll
        for (int k = 0; k < kQrPerThread; ++k) {
            const int a = min(ea[k], n - 1), c = ec[k];
            gv[k] = G[a][c] - G[j][a] * G[j][c];
        }
#pragma unroll
        for (int k = 0; k < kQrPerThread; ++k) {
            const int a = ea[k], c = ec[k];
            if (a > j && a < n && c >= a) G[a][c] = gv[k];
        }
        __syncthreads();
    }
    const float poison = bad ? __builtin_nanf("") : 0.f;
    for (int e = tid; e < n * n; e += 256) {
        const int i = e / n, c = e % n;
        R_out[e] = c >= i ? G[i][c] + poison : 0.f;
    }
    if (tid < n) {
        const int c = tid;
        for (int i = n - 1; i >= 0; --i) {
            float pa[8] = {i == c ? 1.f : 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int k = i + 1;
            for (; k + 8 <= c + 1; k += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) pa[u] -= G[i][k + u] * Ri[k + u][c];
            }
            for (; k <= c; ++k) pa[0] -= G[i][k] * Ri[k][c];
            const float acc = ((pa[0] + pa[1]) + (pa[2] + pa[3])) + ((pa[4] + pa[5]) + (pa[6] + pa[7]));
            Ri[i][c] = i <= c ? acc / G[i][i] : 0.f;
        }
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += 256) Rinv_out[e] = Ri[e / n][e % n] + (e / n <= e % n ? poison : 0.f);
    if (status && tid == 0) {
        status[0] = red[0];
        status[1] = bad ? 1.f : 0.f;
    }
}

// out[n][n] = X^T Y for two tall-skinny matrices [m x n], n <= 64 (the Gram matrices of the QR above and X = -Q^T dQ of its
// backward): K = 7168, M = N = 50 is a shape the GEMM library answers with one 32 x 32 x 256 macro-tile kernel of 48 us; here 64
// rows per block (112 blocks), a 4 x 4 register tile per thread, partial sums in a fixed order, then one small reduction.
// Either operand may be row- or column-major (element (k, i) at k * rs + i * cs): the basis arrives as the transpose of [n][m].
constexpr int kGramRows = 64;
__global__ void __launch_bounds__(256) tall_gram_kernel(const float* __restrict__ X, long long xrs, long long xcs,
                                                        const float* __restrict__ Y, long long yrs, long long ycs,
                                                        float* __restrict__ partial, int m, int n) {
    __shared__ __attribute__((aligned(16))) float Xs[kGramRows][kQrMax + 4], Ys[kGramRows][kQrMax + 4];
    const int tid = threadIdx.x, k0 = blockIdx.x * kGramRows;
    // (all 2 x 16 loads of a thread go out before the first LDS write: one memory round trip per block, not sixteen)
    constexpr int NE = kGramRows * kQrMax / 256;
    float xv[NE], yv[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int e = tid + 256 * u;
        // consecutive threads along the unit-stride direction of the operand
        const int kx = xcs == 1 ? e / kQrMax : e % kGramRows, ix = xcs == 1 ? e % kQrMax : e / kGramRows;
        const bool okx = k0 + kx < m && ix < n;
        xv[u] = X[okx ? (long long)(k0 + kx) * xrs + ix * xcs : 0];
        xv[u] = okx ? xv[u] : 0.f;
        const int ky = ycs == 1 ? e / kQrMax : e % kGramRows, iy = ycs == 1 ? e % kQrMax : e / kGramRows;
        const bool oky = k0 + ky < m && iy < n;
        yv[u] = Y[oky ? (long long)(k0 + ky) * yrs + iy * ycs : 0];
        yv[u] = oky ? yv[u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int e = tid + 256 * u;
        Xs[xcs == 1 ? e / kQrMax : e % kGramRows][xcs == 1 ? e % kQrMax : e / kGramRows] = xv[u];
        Ys[ycs == 1 ? e / kQrMax : e % kGramRows][ycs == 1 ? e % kQrMax : e / kGramRows] = yv[u];
    }
    __syncthreads();
    const int ti = tid >> 4, tj = tid & 15;
    float acc[4][4] = {};
#pragma unroll 8
    for (int k = 0; k < kGramRows; ++k) {
        const float4 a = *reinterpret_cast<const float4*>(&Xs[k][4 * ti]), b = *reinterpret_cast<const float4*>(&Ys[k][4 * tj]);
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    float* dst = partial + (size_t)blockIdx.x * n * n;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (4 * ti + i < n && 4 * tj + j < n) dst[(4 * ti + i) * n + 4 * tj + j] = acc[i][j];
}

// block = 16 consecutive outputs x 16 partial lanes: lane q sums partials q, q + 16, ... (all loads in flight), then a fixed tree
__global__ void __launch_bounds__(256) tall_gram_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int nblocks,
                                                               int nn, float scale) {
    __shared__ float red[16][17];
    const int kk = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + kk;
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (e < nn) {
        for (int b = q; b < nblocks; b += 8 * 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (b + 16 * u < nblocks) p[u] += partial[(size_t)(b + 16 * u) * nn + e];
        }
    }
    red[q][kk] = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    __syncthreads();
    if (q == 0 && e < nn) {
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = red[u][kk];
#pragma unroll
        for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (u < w) t[u] += t[u + w];
        out[e] = t[0] * scale;
    }
}

}  // namespace hfagp

using namespace hfagp;

extern "C" size_t hfagp_tall_gram_workspace_bytes(int32_t m, int32_t n) {
    return (size_t)((m + kGramRows - 1) / kGramRows) * n * n * sizeof(float);
}

extern "C" int hfagp_tall_gram(const float* X, int64_t x_rs, int64_t x_cs, const float* Y, int64_t y_rs, int64_t y_cs, float* workspace,
                               float* out, int32_t m, int32_t n, float scale, void* stream) {
    HFAGP_REQUIRE(X && Y && workspace && out, HFAGP_EBADARG, "tall_gram: null pointer");
    HFAGP_REQUIRE(m >= 1 && n >= 1 && n <= kQrMax, HFAGP_EUNSUPPORTED, "tall_gram: n=%d must be in 1..%d", n, kQrMax);
    const int nblocks = (m + kGramRows - 1) / kGramRows;
    hipStream_t s = (hipStream_t)stream;
    tall_gram_kernel<<<nblocks, 256, 0, s>>>(X, x_rs, x_cs, Y, y_rs, y_cs, workspace, m, n);
    tall_gram_reduce_kernel<<<(n * n + 15) / 16, 256, 0, s>>>(workspace, out, nblocks, n * n, scale);
    return check_launch("tall_gram");
}

extern "C" int hfagp_qr_refine_fwd(const float* gram, float* R, float* Rinv, float* status, int32_t n, void* stream) {
    HFAGP_REQUIRE(gram && R && Rinv, HFAGP_EBADARG, "qr_refine_fwd: null pointer");
    HFAGP_REQUIRE(n >= 1 && n <= kQrMax, HFAGP_EUNSUPPORTED, "qr_refine_fwd: n=%d must be in 1..%d", n, kQrMax);
    qr_refine_kernel<<<1, 256, 0, (hipStream_t)stream>>>(gram, R, Rinv, status, n);
    return check_launch("qr_refine_fwd");
}

extern "C" int hfagp_qr_gram_fwd(const float* gram, const float* top, float* R, float* Rinv, int32_t n, void* stream) {
    HFAGP_REQUIRE(gram && top && R && Rinv, HFAGP_EBADARG, "qr_gram_fwd: null pointer");
    HFAGP_REQUIRE(n >= 1 && n <= kQrMax, HFAGP_EUNSUPPORTED, "qr_gram_fwd: n=%d must be in 1..%d", n, kQrMax);
    qr_gram_kernel<<<1, 256, 0, (hipStream_t)stream>>>(gram, top, R, Rinv, n);
    return check_launch("qr_gram_fwd");
}

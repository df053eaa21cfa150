// Loss side of the fitting step (trainer_rgb.py:84-86, trainer_3dmm.py:51-53, trainer_audio.py:97-99):
//   generated = AdaptiveAvgPool2d(size)(image)   [B][C][H][W] -> [B][C][h][w],  H = f*h, W = f*w
//   l2        = MSELoss(reduction='mean')(real, generated)
// as ONE pass over the image (pooled image + per-block partial sums, reduced in a fixed order -> deterministic) and
// ONE pass for the adjoint d image = g * 2 (pooled - real) / (N f^2).  SURVEY.md section 8f rank 3.  gfx950 only.
#include "common.h"

namespace hfagp {

constexpr int kLossBlocks = 1024;

// thread = one pooled pixel; f x f window read with f row segments (f = 2: one float2 per row)
__global__ void __launch_bounds__(256) pool_mse_fwd_kernel(const float* __restrict__ img, const float* __restrict__ real,
                                                           float* __restrict__ pooled, float* __restrict__ partial,
                                                           long long n, int h, int w, int f) {
    __shared__ float red[4];
    const int W = w * f;
    const float inv = 1.f / (float)(f * f);
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % w), y = (int)((i / w) % h);
        const long long bc = i / ((long long)w * h);
        const float* src = img + (bc * h * f + (long long)y * f) * W + (long long)x * f;
        float s = 0.f;
        if (f == 2) {
            const float2 r0 = *reinterpret_cast<const float2*>(src), r1 = *reinterpret_cast<const float2*>(src + W);
            s = (r0.x + r0.y) + (r1.x + r1.y);
        } else {
            for (int dy = 0; dy < f; ++dy)
                for (int dx = 0; dx < f; ++dx) s += src[(long long)dy * W + dx];
        }
        const float p = s * inv;
        pooled[i] = p;
        const float d = real[i] - p;
        acc += d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// one wave: fixed-order sum of the block partials, scaled by 1/n
__global__ void __launch_bounds__(64) loss_finalize_kernel(const float* __restrict__ partial, float* __restrict__ loss,
                                                           int nblocks, float inv_n) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 64) acc += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (threadIdx.x == 0) loss[0] = acc * inv_n;
}

// thread = one pooled pixel: writes its f x f window of d image
__global__ void __launch_bounds__(256) pool_mse_bwd_kernel(const float* __restrict__ pooled, const float* __restrict__ real,
                                                           const float* __restrict__ g_loss, float* __restrict__ d_img,
                                                           long long n, int h, int w, int f) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int W = w * f;
    const int x = (int)(i % w), y = (int)((i / w) % h);
    const long long bc = i / ((long long)w * h);
    const float v = g_loss[0] * 2.f * (pooled[i] - real[i]) / ((float)n * (float)(f * f));
    float* dst = d_img + (bc * h * f + (long long)y * f) * W + (long long)x * f;
    if (f == 2) {
        *reinterpret_cast<float2*>(dst) = make_float2(v, v);
        *reinterpret_cast<float2*>(dst + W) = make_float2(v, v);
    } else {
        for (int dy = 0; dy < f; ++dy)
            for (int dx = 0; dx < f; ++dx) dst[(long long)dy * W + dx] = v;
    }
}

}  // namespace hfagp

using namespace hfagp;

extern "C" {

size_t hfagp_pool_mse_workspace_bytes(void) { return kLossBlocks * sizeof(float); }

int hfagp_pool_mse_fwd(const float* img, const float* real, float* pooled, float* loss, float* workspace,
                       int32_t BC, int32_t h, int32_t w, int32_t f, void* stream) {
    HFAGP_REQUIRE(img && real && pooled && loss && workspace, HFAGP_EBADARG, "pool_mse_fwd: null pointer");
    HFAGP_REQUIRE(BC > 0 && h > 0 && w > 0 && f >= 1, HFAGP_EBADARG, "pool_mse_fwd: bad dims");
    HFAGP_REQUIRE(f != 2 || (w * f) % 2 == 0, HFAGP_EUNSUPPORTED, "pool_mse_fwd: row pitch");
    const long long n = (long long)BC * h * w;
    long long blocks = (n + 255) / 256;
    if (blocks > kLossBlocks) blocks = kLossBlocks;
    hipStream_t s = (hipStream_t)stream;
    pool_mse_fwd_kernel<<<(unsigned)blocks, 256, 0, s>>>(img, real, pooled, workspace, n, h, w, f);
    loss_finalize_kernel<<<1, 64, 0, s>>>(workspace, loss, (int)blocks, 1.f / (float)n);
    return check_launch("pool_mse_fwd");
}

int hfagp_pool_mse_bwd(const float* pooled, const float* real, const float* g_loss, float* d_img,
                       int32_t BC, int32_t h, int32_t w, int32_t f, void* stream) {
    HFAGP_REQUIRE(pooled && real && g_loss && d_img, HFAGP_EBADARG, "pool_mse_bwd: null pointer");
    HFAGP_REQUIRE(BC > 0 && h > 0 && w > 0 && f >= 1, HFAGP_EBADARG, "pool_mse_bwd: bad dims");
    const long long n = (long long)BC * h * w;
    pool_mse_bwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(pooled, real, g_loss, d_img, n, h, w, f);
    return check_launch("pool_mse_bwd");
}

}  // extern "C"

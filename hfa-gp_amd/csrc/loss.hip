// Loss side of the fitting step (trainer_rgb.py:84-86, trainer_3dmm.py:51-53, trainer_audio.py:97-99):
//   generated = AdaptiveAvgPool2d(size)(image)   [B][C][H][W] -> [B][C][h][w],  H = f*h, W = f*w
//   l2        = MSELoss(reduction='mean')(real, generated)
// as ONE pass over the image (pooled image + per-block partial sums, reduced in a fixed order -> deterministic) and
// ONE pass for the adjoint d image = g * 2 (pooled - real) / (N f^2).  SURVEY.md section 8f rank 3.  gfx950 only.
#include <cmath>
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

// ---------------------------------------------------------------------------------------------------------------
// Adam over MANY parameter tensors in one launch (round 5; torch.optim.Adam's update rule, trainer_rgb.py:58: lr 3e-4, betas
// (0.9, 0.999), eps 1e-8, no weight decay / amsgrad).  When the generator is tuned the step updates 30.7 M + the driver net's
// parameters: 28 bytes of traffic per parameter, i.e. HBM-bound; PyTorch's fused multi-tensor Adam moved 1.9 TB/s here (19 launches,
// 0.55 ms per step).  Tables live in device memory (built once per parameter set by the host): per tensor {p, g, m, v, step, numel}
// as six 64-bit words, per block {tensor, first element}.  `step` (one float per tensor, torch's state layout) is advanced by
// adam_advance_kernel first; the bias corrections are formed in double like torch's.
constexpr int kAdamChunk = 16384;       // elements per block: 4 x 16 B per thread and stream, 4 passes

__global__ void __launch_bounds__(256) adam_advance_kernel(const long long* __restrict__ tensors, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) *reinterpret_cast<float*>(tensors[6 * i + 4]) += 1.f;
}

__global__ void __launch_bounds__(256) adam_update_kernel(const long long* __restrict__ tensors, const int2* __restrict__ chunks,
                                                          double lr, double beta1d, double beta2d, float eps, double ln_b1,
                                                          double ln_b2) {
    __shared__ float bc[2];
    const int2 ch = chunks[blockIdx.x];
    const long long* t = tensors + 6 * (long long)ch.x;
    float* p = reinterpret_cast<float*>(t[0]);
    const float* g = reinterpret_cast<const float*>(t[1]);
    float* m = reinterpret_cast<float*>(t[2]);
    float* v = reinterpret_cast<float*>(t[3]);
    const long long numel = t[5];
    if (threadIdx.x == 0) {
        const double step = (double)*reinterpret_cast<const float*>(t[4]);
        // beta^step = exp(step ln beta), ln beta from the host: the generic double pow() on ONE thread, with the other 255 waiting at
        // the barrier, was most of this kernel whenever the step covers few parameters (84 us for the 110 chunks of driver + basis)
        bc[0] = (float)(lr / (1.0 - exp(step * ln_b1)));                  // lr / bias_correction1
        bc[1] = (float)(1.0 / sqrt(1.0 - exp(step * ln_b2)));             // 1 / sqrt(bias_correction2)
    }
    __syncthreads();
    // (1 - beta in DOUBLE, then rounded: 1.f - 0.999f is 1.3e-5 off the factor torch multiplies with)
    const float step_size = bc[0], rs2 = bc[1], omb1 = (float)(1.0 - beta1d), omb2 = (float)(1.0 - beta2d), beta2 = (float)beta2d;
    const long long e0 = (long long)ch.y, e1 = min(numel, e0 + kAdamChunk);
    auto one = [&](float& pp, float gg, float& mm, float& vv) {
        mm = mm + (gg - mm) * omb1;
        vv = vv * beta2 + gg * gg * omb2;
        pp -= step_size * mm / (sqrtf(vv) * rs2 + eps);
    };
    // (p, m, v 16-byte aligned; the gradient is a slice of the trainer's flat buffer, packed without padding: after the first
    // parameter whose element count is not a multiple of 4 every slice is only 4-byte aligned — global 16-byte loads take that, and
    // demanding 16 sent every later tensor down the scalar loop: 64 dependent rounds per thread, 84 us for driver + basis)
    const bool vec = (((unsigned long long)t[0] | (unsigned long long)t[2] | (unsigned long long)t[3]) & 15ull) == 0 &&
                     ((unsigned long long)t[1] & 3ull) == 0 && (e0 & 3) == 0;
    if (vec) {
        const long long ev = e0 + ((e1 - e0) & ~3ll);            // end of the whole float4s
        for (long long i = e0 + 4 * threadIdx.x; i < ev; i += 1024) {
            float4 pp = *reinterpret_cast<float4*>(p + i), mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
            float4 gg;
            __builtin_memcpy(&gg, g + i, 16);            // (4-byte aligned: the compiler must not assume more)
            one(pp.x, gg.x, mm.x, vv.x); one(pp.y, gg.y, mm.y, vv.y); one(pp.z, gg.z, mm.z, vv.z); one(pp.w, gg.w, mm.w, vv.w);
            *reinterpret_cast<float4*>(p + i) = pp; *reinterpret_cast<float4*>(m + i) = mm; *reinterpret_cast<float4*>(v + i) = vv;
        }
        if (ev + threadIdx.x < e1) { const long long i = ev + threadIdx.x; one(p[i], g[i], m[i], v[i]); }     // ragged tail: <= 3 elements
    } else {
        for (long long i = e0 + threadIdx.x; i < e1; i += 256) one(p[i], g[i], m[i], v[i]);
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

extern "C" int hfagp_adam_step(const void* tensor_table, const void* chunk_table, int32_t ntensors, int32_t nchunks, double lr,
                               double beta1, double beta2, double eps, void* stream) {
    HFAGP_REQUIRE(tensor_table && chunk_table && ntensors >= 1 && nchunks >= 1, HFAGP_EBADARG, "adam_step: empty tables");
    HFAGP_REQUIRE(lr == lr && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0, HFAGP_EBADARG,
                  "adam_step: lr=%g betas=(%g, %g) eps=%g", lr, beta1, beta2, eps);
    hipStream_t s = (hipStream_t)stream;
    adam_advance_kernel<<<(ntensors + 255) / 256, 256, 0, s>>>(static_cast<const long long*>(tensor_table), ntensors);
    adam_update_kernel<<<(unsigned)nchunks, 256, 0, s>>>(static_cast<const long long*>(tensor_table), static_cast<const int2*>(chunk_table),
                                                         lr, beta1, beta2, (float)eps, log(beta1), log(beta2));
    return check_launch("adam_step");
}

extern "C" int32_t hfagp_adam_chunk(void) { return kAdamChunk; }

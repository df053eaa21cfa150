// Developer micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate on this box (registers only, random operands),
// the practical ceiling of the split-bf16 conv kernels.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ void __launch_bounds__(256) k(const uint4* in, float* out, int iters) {
    uint4 ua = in[threadIdx.x], ub = in[threadIdx.x + 256];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc[i], 0, 0, 0);
        ua.x ^= it;                                            // operands change: realistic toggle rate
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    uint4* in; float* out; const int iters = 40000;
    hipMalloc(&in, 512 * 16); hipMalloc(&out, 256 * 3 * 256 * 4);
    unsigned h[2048]; for (int i = 0; i < 2048; ++i) h[i] = ((unsigned)rand() & 0x3fff3fffu) | 0x3c003c00u;   // finite bf16 pairs
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves = 1; waves <= 3; ++waves)
        for (int rep = 0; rep < 2; ++rep) {
            const int blocks = 256 * waves;
            hipEventRecord(e0); k<4><<<blocks, 256>>>(in, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fl = (double)blocks * 4 * iters * 4 * 32768.0;
            printf("%d wave(s)/SIMD: %.3f ms, %.0f TFLOP/s f16 (%.0f TFLOP/s as bf16x3, %.0f as bf16x6)\n", waves, ms,
                   fl / ms / 1e9, fl / ms / 3e9, fl / ms / 6e9);
        }
    return 0;
}

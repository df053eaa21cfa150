// Developer micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate on this box (registers only, random operands).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k(const float* in, float* out, int iters) {
    float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-7f;
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *in, *out; const int blocks = 256 * 3, iters = 20000;
    hipMalloc(&in, 512 * 4); hipMalloc(&out, blocks * 256 * 4);
    float h[512]; for (int i = 0; i < 512; ++i) h[i] = (float)rand() / RAND_MAX * 2 - 1;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); k<4><<<blocks, 256>>>(in, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)blocks * 4 * iters * 4 * 4096.0;
        printf("blocks=%d (3 waves/SIMD) NACC=4: %.3f ms, %.1f TFLOP/s\n", blocks, ms, fl / ms / 1e9);
    }
    hipEventRecord(e0); k<4><<<256, 256>>>(in, out, iters * 3); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("blocks=256 (1 wave/SIMD): %.3f ms, %.1f TFLOP/s\n", ms, 256.0 * 4 * iters * 3 * 4 * 4096.0 / ms / 1e9);
    return 0;
}

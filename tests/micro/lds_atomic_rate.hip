// Micro-benchmark: throughput of LDS fp32 adds on gfx950 (256 threads / block, 2 blocks per CU):
//   mode 0  ds_add_f32 (no return), 64 consecutive floats per wave instruction (conflict-free), rotating rows
//   mode 1  the same, both half-waves on the SAME 32 floats (2-way same-address collision, the scatter's worst case)
//   mode 2  plain read-modify-write (ds_read_b32 + v_add + ds_write_b32), no atomicity, for reference
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic_rate.hip -o lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(256, 2) k(float* out, int iters, int mode) {
    __shared__ float win[64 * 64];                    // 16 KB: 64 rows of 64 floats
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) win[i] = 0.f;
    __syncthreads();
    const int col = mode == 1 ? (lane & 31) : lane;
    float v = 1.0f + lane;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int row = (i * 16 + u * 5 + wave * 16) & 63;
            if (mode == 2) win[row * 64 + col] += v;
            else atomicAdd(&win[row * 64 + col], v);
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = win[threadIdx.x * 64 + threadIdx.x];
}

int main() {
    float* out; hipMalloc(&out, 4096 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 2 * 4, iters = 2048;
    const char* names[] = {"ds_add_f32, 64 distinct floats / instr", "ds_add_f32, half-waves collide (32 floats)", "ds_read + add + ds_write (not atomic)"};
    for (int mode = 0; mode < 3; ++mode) {
        k<<<blocks, 256>>>(out, 16, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<blocks, 256>>>(out, iters, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double lanes = (double)blocks * 256 * iters * 16;
        printf("%-48s: %.3f ms, %.1f G lane-adds/s = %.2f lanes/clk/CU at 2.4 GHz (%.2f G 128-B lines/s)\n", names[mode], ms,
               lanes / ms * 1e-6, lanes / ms * 1e-6 / 256 / 2.4, lanes / 32 / ms * 1e-6);
    }
    return 0;
}

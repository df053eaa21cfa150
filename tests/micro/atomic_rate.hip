// Micro-benchmark: throughput of 128-byte fp32 line atomics (global_atomic_add_f32, 32 lanes per line, two lines per
// wave instruction) on gfx950 for different address patterns.  hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) k(float* buf, const unsigned* lines, long long per_wave, unsigned nlines_mask,
                                        int mode, int reps) {
    const int lane = threadIdx.x & 63, c = lane & 31, hf = lane >> 5;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (long long i = 0; i < per_wave; ++i) {
        const long long e = (wave * per_wave + i) * 2 + hf;
        unsigned line;
        if (mode == 0) line = (unsigned)(e * 2654435761ull >> 7) & nlines_mask;          // pseudo-random lines
        else if (mode == 1) line = (unsigned)e & nlines_mask;                            // sequential lines
        else if (mode == 2) line = ((unsigned)(wave * 64) + (unsigned)(i & 31) * 2 + hf) & nlines_mask;   // per-wave window of 64 lines
        else line = lines[e & 0xfffff] & nlines_mask;
        for (int r = 0; r < reps; ++r) unsafeAtomicAdd(buf + (size_t)line * 32 + c, 1.0f);
    }
}

int main() {
    const unsigned nlines = 1u << 18;      // 32 MB
    float* buf; hipMalloc(&buf, (size_t)nlines * 128); hipMemset(buf, 0, (size_t)nlines * 128);
    unsigned* lines; hipMalloc(&lines, 4u << 20);
    const int blocks = 256 * 8;
    const long long per_wave = 2048;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"random lines in 32 MB", "sequential lines", "64-line window per wave (L2-resident 4 MB total)"};
    for (int mode = 0; mode < 3; ++mode)
        for (int reps = 1; reps <= 4; reps *= 4) {
            k<<<blocks, 256>>>(buf, lines, 64, nlines - 1, mode, reps);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            k<<<blocks, 256>>>(buf, lines, per_wave, nlines - 1, mode, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double n = (double)blocks * 4 * per_wave * 2 * reps;      // line atomics
            printf("%-50s reps=%d: %.3f ms, %.2f G line-atomics/s = %.1f G lane-atomics/s\n", names[mode], reps, ms,
                   n / ms * 1e-6, n * 32 / ms * 1e-6);
        }
    return 0;
}

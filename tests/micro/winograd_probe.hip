// Developer micro-benchmark: an UPPER BOUND on a fused Winograd F(2x2, 3x3) version of the 9-tap conv
// GEMM on split fp16 operands (f16x3), layer 256 -> 256 @ 256^2, B = 8 — the layer the direct kernel runs at ~390 TFLOP/s
// (1.58 ms).  The probe runs ONLY the frequency-domain stage of such a kernel, with its real operand traffic:
//   block = 4 waves, 32 Winograd tiles (8 x 16 output pixels, as the direct kernel) x 128 output channels;
//   16 frequencies x [32 tiles x 128 ch] fp32 accumulators = 256 registers per lane (so ONE wave per SIMD: 512-register budget);
//   per 16-channel K chunk and frequency: A fragment (hi, lo parts) from LDS, B fragment (hi, lo parts of the TRANSFORMED
//   weight, 16 / 9 the size of the direct image) straight from L2 as in modconv_bf16_kernel, 3 MFMAs;
//   one barrier per chunk.  NOT included (all of it extra work for the real kernel): reading the input patch, the input
//   transform B^T d B (32 adds per tile and channel), splitting 2.8x as many values into fp16 parts and writing them to LDS,
//   the output transform A^T m A, the epilogue and the stores.
// hipcc --offload-arch=gfx950 -O3 -o winograd_probe winograd_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int NF = 16, CIN = 256, COUT = 256, CK = 16, APITCH = 48;
constexpr int A_FREQ = 32 * APITCH;            // one (part, frequency) image: 32 tiles x (16 halfs + pad)
constexpr int A_BUF = 2 * NF * A_FREQ;         // 2 parts

template <bool LOADB>
__global__ void __launch_bounds__(256, 1) wino_stage(const uint4* __restrict__ wb, float* __restrict__ out, int tiles_per_block) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
    for (int i = tid; i < 2 * A_BUF / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003800u + (i * 2654435761u >> 20 & 0x03ff03ffu);
    __syncthreads();
    const int cq8 = CIN / 8, part_stride = NF * cq8 * COUT;
    const int tn = blockIdx.x & 1;                                  // 128-channel tile of the 256 output channels
    const unsigned bth = (unsigned)(h * COUT + tn * 128 + wave * 32 + l31);
    const int apos = l31 * APITCH + 16 * h;
    float sum = 0.f;
    for (int t = 0; t < tiles_per_block; ++t) {
        f32x16 acc[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
        for (int c = 0; c < CIN / CK; ++c) {
            __syncthreads();                                        // (the real kernel publishes the transformed patch here)
            const char* Ab = lds + (c & 1) * A_BUF;
            u32x4 bq[2][2];
            auto loadb = [&](int f, int slot) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (LOADB) bq[slot][q] = reinterpret_cast<const u32x4*>(wb)[(size_t)q * part_stride + (size_t)(f * cq8 + c * 2) * COUT + bth];
                    else bq[slot][q] = u32x4{0x3c003c00u + (unsigned)f, 0x3c003c00u, 0x38003c00u, 0x3c003a00u + (unsigned)c};
                }
            };
            loadb(0, 0);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (f + 1 < NF) loadb(f + 1, (f + 1) & 1);
                const u32x4 a_hi = *reinterpret_cast<const u32x4*>(Ab + (0 * NF + f) * A_FREQ + apos);
                const u32x4 a_lo = *reinterpret_cast<const u32x4*>(Ab + (1 * NF + f) * A_FREQ + apos);
                const f16x8 ah = __builtin_bit_cast(f16x8, a_hi), al = __builtin_bit_cast(f16x8, a_lo);
                const f16x8 bh = __builtin_bit_cast(f16x8, bq[f & 1][0]), bl = __builtin_bit_cast(f16x8, bq[f & 1][1]);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[f], 0, 0, 0);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[f], 0, 0, 0);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[f], 0, 0, 0);
            }
        }
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[f][r];
    }
    out[blockIdx.x * 256 + tid] = sum;
}

int main() {
    const int B = 8, H = 256, W = 256;
    const long long tiles = (long long)B * H * W / 128 * (COUT / 128);         // block tiles of the layer (8192)
    const size_t wbytes = (size_t)2 * NF * (CIN / 8) * COUT * 16;              // transformed split weight image
    uint4* wb; float* out;
    hipMalloc(&wb, wbytes); hipMalloc(&out, 4096 * 256 * 4);
    std::vector<unsigned> hw(wbytes / 4);
    for (auto& v : hw) v = ((unsigned)rand() & 0x03ff03ffu) | 0x38003800u;
    hipMemcpy(wb, hw.data(), wbytes, hipMemcpyHostToDevice);
    const size_t ldsb = 2 * A_BUF;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_stage<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_stage<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flops_direct = 2.0 * B * H * W * CIN * COUT * 9;
    printf("layer 256->256 @256^2, B=8: %.0f GFLOP as a direct conv; Winograd weight image %.2f MB (direct: %.2f MB; L2 per XCD: 4 MB)\n",
           flops_direct / 1e9, wbytes / 1048576.0, wbytes * 9.0 / 16 / 1048576.0);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            const int blocks = 1024, per = (int)(tiles / blocks);
            hipEventRecord(e0);
            if (mode == 0) wino_stage<true><<<blocks, 256, ldsb>>>(wb, out, per); else wino_stage<false><<<blocks, 256, ldsb>>>(wb, out, per);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double mf = (double)tiles * 4 * (CIN / CK) * NF * 3 * 65536.0;   // MFMA flops actually issued (incl. the 3x split)
            printf("%s: %.3f ms  -> %.0f TFLOP/s direct-conv equivalent (direct kernel: ~390), MFMA rate %.0f TFLOP/s 16-bit\n",
                   mode == 0 ? "frequency stage, B fragments from L2 " : "frequency stage, B fragments synthetic", ms,
                   flops_direct / ms / 1e9, mf / ms / 1e9);
        }
    printf("hipGetLastError: %s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}

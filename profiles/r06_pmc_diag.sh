#!/usr/bin/env bash
# Run ON THE GPU BOX (round 6): PMC passes (own runs, kernel-trace only) of the kernels this round built or asked about:
#   upfir_lean_kernel<4>          first SR layer, 32 -> 256 @128^2 -> 256^2, B = 32
#   upconv_bf16_kernel<4, 4, 0>   256 -> 128 @256^2 -> 513^2, B = 32 (stacked rows + fringe tiles, two 4-wave blocks per CU)
#   wgrad_up_bf16_kernel          256 -> 128 @256^2, B = 2 (VERDICT r5 #4a: "commit its PMC beside wgrad_256_256_at256.txt")
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06_pmc
UPFIR_LAYERS="128,32,256" UPFIR_NOSYN=1 bash profiles/pmc_kernel.sh "upfir_lean_kernel" tools/dev/bench_upfir.py 32 > gpurun_out/r06_pmc/upfir_lean_32_256_in128.txt 2>&1
bash profiles/pmc_kernel.sh "upconv_bf16_kernel" tools/dev/bench_conv.py 32 256 256 128 2 0 10 f16x3 > gpurun_out/r06_pmc/upconv_256_128_in256.txt 2>&1
bash profiles/pmc_kernel.sh "wgrad_up_bf16_kernel" tools/dev/bench_wgrad.py 2 256 256 128 up bf16x3 > gpurun_out/r06_pmc/wgrad_up_256_128_at256.txt 2>&1
bash profiles/pmc_kernel.sh "wgrad_bf16_kernel" tools/dev/bench_wgrad.py 2 256 256 256 bf16x3 > gpurun_out/r06_pmc/wgrad_256_256_at256.txt 2>&1
head -40 gpurun_out/r06_pmc/*.txt

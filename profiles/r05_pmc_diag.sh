#!/usr/bin/env bash
# Run ON THE GPU BOX (via gpurun): SQ / TCC / traffic counters of the merged up-conv GEMM beside the 9-tap GEMM on the SAME operands
# (256 -> 128 channels, 256^2 input, B = 8, f16x3) and of the 3x3 weight-gradient GEMM.  Separate --pmc passes
# per counter group, kernel-trace only (profiles/pmc_kernel.sh).  Output: gpurun_out/r05_pmc/*.txt
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
o="$R/gpurun_out/r05_pmc"; mkdir -p "$o"
bash "$R/profiles/pmc_kernel.sh" upconv_bf16_kernel  tools/dev/bench_conv.py 8 256 256 128 2 0 10 f16x3 > "$o/upconv_256_128_in256.txt" 2>&1
bash "$R/profiles/pmc_kernel.sh" modconv_bf16_kernel tools/dev/bench_conv.py 8 256 256 128 1 0 10 f16x3 > "$o/conv3x3_256_128_at256.txt" 2>&1
bash "$R/profiles/pmc_kernel.sh" wgrad_bf16_kernel   tools/dev/bench_wgrad.py 2 256 256 256 bf16x3      > "$o/wgrad_256_256_at256.txt" 2>&1
for f in "$o"/*.txt; do echo "== $f"; cat "$f"; done
python "$R/tools/dev/bench_conv.py" 8 256 256 128 2 0 10 f16x3
python "$R/tools/dev/bench_conv.py" 8 256 256 128 1 0 10 f16x3
python "$R/tools/dev/bench_wgrad.py" 2 256 256 256 bf16x3
rm -rf "$R"/gpurun_out/pmc_[0-9]*

#!/usr/bin/env bash
# Round 6: merged up-conv GEMM, the 8-wave block (N = 128, one block per CU) against two independent 4-wave blocks per CU (N = 64).
# usage (GPU box): bash tools/dev/up_waves_ab.sh > gpurun_out/up_waves_ab.log
cd "$(dirname "$0")/../.."
for B in 32 8 2 1; do
  for L in "256 256 128" "128 256 128" "64 512 256" "32 512 512" "16 512 512"; do
    set -- $L
    for wv in 8 4; do
      echo -n "waves=$wv "
      HFAGP_DEV_UP_WAVES=$wv python tools/dev/bench_conv.py $B $1 $2 $3 2 0 20 f16x3 2>&1 | tail -1
    done
  done
done

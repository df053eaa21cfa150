#!/usr/bin/env bash
# Round 6: merged up-conv GEMM, rounds 2-5's per-sample tiling (HFAGP_DEV_UP_LEGACY_TILES=1) against stacked rows + fringe tiles.
# usage (GPU box): bash tools/dev/up_tiles_ab.sh > gpurun_out/up_tiles_ab.log
cd "$(dirname "$0")/../.."
for B in 32 1; do
  for L in "256 256 128" "128 256 128" "64 512 256" "32 512 512" "16 512 512" "8 512 512" "4 512 512"; do
    set -- $L
    for legacy in 1 0; do
      echo -n "legacy=$legacy "
      HFAGP_DEV_UP_LEGACY_TILES=$legacy python tools/dev/bench_conv.py $B $1 $2 $3 2 0 20 f16x3 2>&1 | tail -1
    done
  done
done

#!/usr/bin/env bash
# Round 6: first super-resolution layer (32 -> 256 @128^2 -> 256^2), strip kernel (HFAGP_DEV_FIR_LEAN=0) against the streaming kernel
# (upfir_lean.hip) and the two-kernel form, by batch; then the streaming kernel by segment count.  usage (GPU box): bash tools/dev/upfir_lean_ab.sh
cd "$(dirname "$0")/../.."
export UPFIR_LAYERS="128,32,256" HFAGP_DEV_FIR_MIN_BLOCKS=1 UPFIR_NOSYN=1
for B in 32 16 8 4 2 1; do
  for lean in 0 1; do
    echo -n "B=$B lean=$lean "
    HFAGP_DEV_FIR_LEAN=$lean python tools/dev/bench_upfir.py $B 2>&1 | grep -- "->" | cut -c1-220
  done
done
echo -n "B=32 "; python tools/dev/bench_upfir.py 32 f16x3 1,2,3,4,6,8 2>&1 | grep -- "->" | cut -c1-400
echo -n "B=8 ";  python tools/dev/bench_upfir.py 8 f16x3 1,2,4,8 2>&1 | grep -- "->" | cut -c1-400

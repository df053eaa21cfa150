#!/usr/bin/env bash
# Round 6: conv GEMMs with the XCD-contiguous block mapping (HFAGP_DEV_XCD_REMAP=1) against the plain one, per layer and whole step.
cd "$(dirname "$0")/../.."
for L in "32 256 256 256 0" "32 512 128 128 0" "32 64 512 512 0" "32 256 256 128 2" "32 128 256 128 0"; do
  set -- $L
  for x in 0 1; do
    echo -n "xcd=$x "
    HFAGP_DEV_XCD_REMAP=$x python tools/dev/bench_conv.py $1 $2 $3 $4 $5 0 20 f16x3 2>&1 | tail -1 | cut -c1-110
  done
done
args="--no-cpu-baseline --no-train --no-sweep --no-fp32-leg --no-f16-leg --audio-frames 0"
for i in 1 2; do for x in 0 1; do echo -n "xcd=$x "; HFAGP_DEV_XCD_REMAP=$x python bench.py $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; done; done

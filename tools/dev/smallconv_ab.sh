#!/usr/bin/env bash
# Round 6: 3x3 conv at 4^2 ... 16^2: smallconv_kernel (ksplit 0 = library's choice) against the staged kernel (ksplit forced), by batch.
cd "$(dirname "$0")/../.."
for B in 32 16 8 4 2; do
  for H in 16 8 4; do
    for ks in 0 1 2 4; do
      echo -n "ks=$ks "
      python tools/dev/bench_conv.py $B $H 512 512 0 $ks 20 f16x3 2>&1 | tail -1 | cut -c1-100
    done
  done
done

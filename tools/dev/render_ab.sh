#!/usr/bin/env bash
# Round 6: the headline leg on ONE box, rounds 2-5's up-conv tiling / 8-wave block (developer switches) against the default, alternating.
cd "$(dirname "$0")/../.."
args="--no-cpu-baseline --no-train --no-sweep --no-fp32-leg --no-f16-leg --audio-frames 0"
for i in 1 2 3; do
  for old in 1 0; do
    if [ $old = 1 ]; then export HFAGP_DEV_UP_LEGACY_TILES=1 HFAGP_DEV_UP_WAVES=8; else unset HFAGP_DEV_UP_LEGACY_TILES HFAGP_DEV_UP_WAVES; fi
    echo -n "old_up_tiling=$old "
    python bench.py $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('up_layers_ms_per_step'))"
  done
done

#!/usr/bin/env bash
# Run ON THE GPU BOX: kernel trace of the fitting step (tests/bench_train.py B iters); per-kernel ms per step.
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
tag="${1:-train}"; B="${2:-2}"; iters="${3:-10}"; mode="${4:-3dmm}"
out="/tmp/prof_$tag"; rm -rf "$out"; mkdir -p "$out"      # raw traces stay on the box (gpurun copies back <= 64 MiB)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- python "$R/tests/bench_train.py" "$B" "$iters" "$mode" > "$out/trace.log" 2>&1
python - "$out" "$iters" <<'PY'
import csv, glob, os, sys
out, iters = sys.argv[1], int(sys.argv[2]) + 2
f = sorted(glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True))[0]
tot = 0
for r in csv.DictReader(open(f)):
    ms = float(r["TotalDurationNs"]) / 1e6 / iters
    tot += ms
    if ms > 0.015:
        print(f"{r['Name'].split('(')[0][:70]:70s} calls/step {int(r['Calls'])/iters:6.1f}  ms/step {ms:7.3f}  avg_us {float(r['AverageNs'])/1e3:8.1f}")
print("total kernel ms/step", tot)
PY
tail -1 "$out/trace.log"

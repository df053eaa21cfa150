#!/usr/bin/env bash
# Run ON THE GPU BOX: kernel trace of the headline render leg only (B=8, preset precision); per-kernel ms per step.
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
tag="${1:-render}"; steps="${2:-20}"; shift 2 || true
out="/tmp/prof_$tag"; rm -rf "$out"; mkdir -p "$out"      # raw traces stay on the box (gpurun copies back <= 64 MiB)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- python "$R/bench.py" --steps "$steps" --warmup 3 --no-cpu-baseline --no-train --no-sweep --no-fp32-leg --no-f16-leg --audio-frames 0 "$@" > "$out/trace.log" 2>&1
python - "$out" "$steps" <<'PY'
import csv, glob, os, sys
out, iters = sys.argv[1], 2 * (int(sys.argv[2]) + 3)      # bench.py runs the headline leg twice (plain, then with events)
f = sorted(glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True))[0]
tot = 0
for r in csv.DictReader(open(f)):
    ms = float(r["TotalDurationNs"]) / 1e6 / iters
    tot += ms
    if ms > 0.02:
        print(f"{r['Name'].split('(')[0][:70]:70s} calls/step {int(r['Calls'])/iters:6.1f}  ms/step {ms:7.3f}  avg_us {float(r['AverageNs'])/1e3:8.1f}")
print("total kernel ms/step", tot)
PY
grep '^{' "$out/trace.log" | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"

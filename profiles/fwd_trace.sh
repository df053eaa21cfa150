#!/usr/bin/env bash
# Run ON THE GPU BOX: forward-only kernel trace of the render leg; prints per-kernel ms per step.
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
tag="${1:-fwd}"
out="$R/gpurun_out/prof_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-train > "$out/trace.log" 2>&1
python - "$out" <<'PY'
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "trace", "**", "*kernel_stats.csv"), recursive=True))[0]
tot = 0
for r in csv.DictReader(open(f)):
    ms = float(r["TotalDurationNs"]) / 1e6 / 13
    tot += ms
    if ms > 0.01:
        print(f"{r['Name'].split('(')[0][:70]:70s} calls/step {int(r['Calls'])/13:6.1f}  ms/step {ms:7.3f}  avg_us {float(r['AverageNs'])/1e3:8.1f}")
print("total kernel ms/step", tot)
PY
grep '^{' "$out/trace.log" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"

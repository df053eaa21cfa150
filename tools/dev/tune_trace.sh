#!/usr/bin/env bash
# Run ON THE GPU BOX: kernel trace of the fitting step with the generator being tuned (tools/dev/bench_tune.py).
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
out="/tmp/prof_tune"; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- python "$R/tools/dev/bench_tune.py" > "$out/trace.log" 2>&1
tail -2 "$out/trace.log"
python - "$out" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
f = sorted(glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:32]:
    print(f"{r['Name'].split('(')[0][:70]:70s} calls {int(r['Calls']):5d}  total_ms {float(r['TotalDurationNs'])/1e6:8.2f}  avg_us {float(r['AverageNs'])/1e3:8.1f}  {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY

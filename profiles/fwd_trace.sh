#!/usr/bin/env bash
# Run ON THE GPU BOX: forward-only kernel trace of the render leg; prints per-kernel ms per step and the
# per-launch durations of the last step.  usage: fwd_trace.sh <tag> [bench args]
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
tag="${1:-fwd}"; shift || true
out="$R/gpurun_out/prof_$tag"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-train --no-fp32-leg "$@" > "$out/trace.log" 2>&1
python - "$out" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
f = sorted(glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True))[0]
tot = 0
for r in csv.DictReader(open(f)):
    ms = float(r["TotalDurationNs"]) / 1e6 / 13
    tot += ms
    if ms > 0.01:
        print(f"{r['Name'].split('(')[0][:70]:70s} calls/step {int(r['Calls'])/13:6.1f}  ms/step {ms:7.3f}  avg_us {float(r['AverageNs'])/1e3:8.1f}")
print("total kernel ms/step", tot)
f = sorted(glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "raymarch_kernel" in r["Kernel_Name"]]
print("--- launches of the last full step (> 15 us)")
for r in rows[idx[-2]:idx[-1]]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d > 15:
        n = r["Kernel_Name"].split("(")[0].replace("void hfagp::", "").replace("hfagp::", "")[:44]
        print(f"{n:44s} {d:8.1f} us  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}")
PY
grep '^{' "$out/trace.log" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'fps', d['value'])"

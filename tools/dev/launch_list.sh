#!/usr/bin/env bash
# Developer (GPU box): per-launch durations (us) of ONE synthesis call at batch $1, in launch order.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
B="${1:-1}"
out=/tmp/prof_launches; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out" -o t -- python "$R/tools/dev/gpu_dev_streams.py" "$B" 1 > "$out/log.txt" 2>&1
python - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last complete synthesis: from the last style_batch_kernel pair backwards
idx = [i for i, n in enumerate(names) if "style_batch_kernel" in n]
start = idx[-2]
prev_end = None
tot = 0.0
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  {r['Kernel_Name'].split('(')[0][:60]}  grid {r.get('Grid_Size', '')} wg {r.get('Workgroup_Size', '')}")
    prev_end = e
    tot += (e - s) / 1e3
print("sum", tot)
PY

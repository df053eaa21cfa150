#!/usr/bin/env bash
# Developer (GPU box): per-launch durations (us) of ONE steady-state fitting step, in launch order.
# usage: fit_launch_list.sh B mode(3dmm|rgb)
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
B="${1:-2}"; mode="${2:-3dmm}"
out=/tmp/prof_fit_list; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out" -o t -- python "$R/tools/dev/bench_train.py" "$B" 10 "$mode" > "$out/log.txt" 2>&1
tail -1 "$out/log.txt"
python - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "raymarch_bwd_cols_kernel" in r["Kernel_Name"] or "raymarch_bwd_tiles_kernel" in r["Kernel_Name"]]
lo, hi = marks[-2], marks[-1]
prev_end, tot = None, 0.0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  {r['Kernel_Name'].split('(')[0][:90]}")
    prev_end = e
    tot += (e - s) / 1e3
span = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3
print(f"sum of kernel us {tot:.1f}   wall us {span:.1f}   launches {hi - lo}")
PY

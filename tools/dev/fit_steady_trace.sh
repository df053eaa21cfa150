#!/usr/bin/env bash
# Developer (GPU box): per-kernel ms per STEADY-STATE fitting step (the last 5 steps of tools/dev/bench_train.py; MIOpen's
# find-mode launches of the first steps are left out).  usage: fit_steady_trace.sh B mode(3dmm|rgb)
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
B="${1:-2}"; mode="${2:-rgb}"
out=/tmp/prof_fit_steady; rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out" -o t -- python "$R/tools/dev/bench_train.py" "$B" 12 "$mode" > "$out/log.txt" 2>&1
tail -1 "$out/log.txt"
python - "$out" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "raymarch_bwd_cols_kernel" in r["Kernel_Name"] or "raymarch_bwd_tiles_kernel" in r["Kernel_Name"]]
lo, hi, n = marks[-6], marks[-1], 5
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows[lo:hi]:
    k = r["Kernel_Name"].split("(")[0][:80]
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
span = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e6 / n
tot = 0.0
for k, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += ms / n
    if ms / n > 0.02:
        print(f"{k:80s} calls/step {c / n:6.1f}  ms/step {ms / n:7.3f}")
print(f"kernel ms/step {tot:.3f}   wall ms/step {span:.3f}")
PY

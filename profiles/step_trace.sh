#!/usr/bin/env bash
# Run ON THE GPU BOX (via gpurun):  bash profiles/step_trace.sh <tag> [B] [iters]
# rocprofv3 --kernel-trace --stats of the fitting step (tools/dev/bench_train.py) in all four regimes — 3DMM / RGB driven,
# generator frozen / tuned — and for each: the per-kernel table (ms per step) and the per-launch list of the LAST step.
# Output: gpurun_out/<tag>_step_<mode>_<frozen|tuned>.txt (copy the ones to be judged into profiles/).
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
tag="${1:-r05}"; B="${2:-2}"; iters="${3:-8}"; regimes="${4:-3dmm:tuned rgb:tuned}"
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
for reg in $regimes; do
  mode="${reg%%:*}"; tun="${reg##*:}"
  out="/tmp/prof_${tag}_${mode}_${tun}"; rm -rf "$out"; mkdir -p "$out"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o t -- python "$R/tools/dev/bench_train.py" "$B" "$iters" "$mode" "$tun" > "$out/log.txt" 2>&1
  python - "$out" "$iters" "$mode" "$tun" "$B" > "$R/gpurun_out/${tag}_step_${mode}_${tun}.txt" <<'PY'
import csv, glob, sys
out, iters, mode, tun, B = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
print(f"# fitting step, {mode}-driven, generator {tun}, B = {B}: rocprofv3 --kernel-trace --stats over tools/dev/bench_train.py "
      f"({iters} timed + 2 warm-up steps)")
print("# " + open(out + "/log.txt").read().strip().splitlines()[-1])
f = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "raymarch_bwd_cols_kernel" in r["Kernel_Name"] or "raymarch_bwd_rows_kernel" in r["Kernel_Name"]
         or ("raymarch_bwd_tiles_kernel" in r["Kernel_Name"] and "false>" not in r["Kernel_Name"].split("(")[0][-8:])]
# one mark per step: the d-planes scatter / gather of the ray-march backward
steps = []
for i in marks:
    if not steps or i - steps[-1] > 20:
        steps.append(i)
lo, hi = steps[-2], steps[-1]
def short(n):
    n = n.split("(")[0]
    for a, b in (("void ", ""), ("hfagp::", ""), ("at::native::", "at::")):
        n = n.replace(a, b)
    return n[:84]
agg = {}
for r in rows[lo:hi]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(short(r["Kernel_Name"]), [0, 0.0]); a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
span = (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3
print(f"# last step: {hi - lo} launches, sum of kernel time {tot / 1e3:.3f} ms, wall {span / 1e3:.3f} ms\n")
print("## per kernel (last step)")
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:84s} launches {n:4d}  ms/step {us / 1e3:7.3f}  avg_us {us / n:8.1f}  {100 * us / tot:5.1f}%")
print("\n## per launch (last step, launch order)")
prev = None
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  grid {r.get('Grid_Size', '?'):>9s}  {short(r['Kernel_Name'])}")
    prev = e
PY
  head -40 "$R/gpurun_out/${tag}_step_${mode}_${tun}.txt"
done

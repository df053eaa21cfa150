#!/usr/bin/env bash
# Developer (GPU box): first contact of the sort + gather ray-march backward: parity against the scatter kernels, then a kernel trace.
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_backward.py -x -q -k "raymarch_bwd" 2>&1 | tail -5
timeout 300 python tools/dev/bench_raybwd.py 2 5 both 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tail -6
timeout 300 python tools/dev/bench_raybwd.py 2 5 both dec 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tail -10
cd /tmp && export TMPDIR=/tmp
for mode in "rows" "rows dec"; do
  out="/tmp/rows_${mode// /_}"; rm -rf "$out"; mkdir -p "$out"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o t -- python "$R/tools/dev/bench_raybwd.py" 2 10 $mode > "$out/log.txt" 2>&1
  echo "== $mode"; tail -1 "$out/log.txt"
  python - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "raymarch" in r["Name"] or "mirror" in r["Name"]:
            print(f'   {r["Name"][:80]:80s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:9.1f}')
PY
done

#!/usr/bin/env bash
# Developer (GPU box): per-kernel times of the ray-march backward, column variant vs tile variant, with and without atomics.
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
for cfg in "cols:" "tiles:HFAGP_DEV_NO_COLS=1" "cols_noatomics:HFAGP_LIB_PATH=$R/hfa-gp_amd/libhfagp_abl_noatomics.so" "tiles_noatomics:HFAGP_DEV_NO_COLS=1 HFAGP_LIB_PATH=$R/hfa-gp_amd/libhfagp_abl_noatomics.so"; do
  name="${cfg%%:*}"; envs="${cfg#*:}"
  out="$R/gpurun_out/raybwd_$name"; rm -rf "$out"; mkdir -p "$out"
  env $envs rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o t -- python "$R/tools/dev/bench_raybwd.py" ${1:-2} 10 > "$out/log.txt" 2>&1
  echo "== $name"; tail -1 "$out/log.txt"
  python - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "raymarch" in r["Name"] or "mirror" in r["Name"]:
            print(f'   {r["Name"][:70]:70s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:9.1f}')
PY
done

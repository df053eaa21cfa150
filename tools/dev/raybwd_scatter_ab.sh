#!/usr/bin/env bash
# Developer (GPU box): the stand-alone scatter kernel of the ray-march backward with its phases compiled out (variant libraries
# libhfagp_abl_rb_<name>.so built by hand from patched copies of csrc/raymarch_bwd.hip; timing only).
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
for name in base ${VARIANTS:-noxy nowalk neither}; do
  lib="$R/hfa-gp_amd/libhfagp_hip.so"; [ $name != base ] && lib="$R/hfa-gp_amd/libhfagp_abl_rb_$name.so"
  [ -f "$lib" ] || continue
  out="/tmp/rbs_$name"; rm -rf "$out"; mkdir -p "$out"
  HFAGP_LIB_PATH=$lib rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o t -- python "$R/tools/dev/bench_raybwd.py" ${1:-2} 10 2 > "$out/log.txt" 2>&1
  echo "== $name"; tail -1 "$out/log.txt"
  python - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "raymarch_bwd" in r["Name"]:
            print(f'   {r["Name"][:70]:70s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:9.1f}')
PY
done

R="${GRAFT_REPO_ROOT:-/root/repo}"
cd $R
HFAGP_DEV_PG_WIDE=1 python tools/dev/bench_raybwd.py 2 5 both dec 2>&1 | grep -v "^RCCL\|amdgpu.ids" | tail -7
cd /tmp && export TMPDIR=/tmp
run() {   # name, mode, env...
  name=$1; mode=$2; shift 2
  out="/tmp/rows_$name"; rm -rf "$out"; mkdir -p "$out"
  env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o t -- python "$R/tools/dev/bench_raybwd.py" 2 10 $mode > "$out/log.txt" 2>&1
  echo "== $name ($*)"; grep "raymarch_bwd B" "$out/log.txt" | tail -1
  python - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "raymarch_bwd_df" in r["Name"] or "tiles" in r["Name"]:
            print(f'   {r["Name"][:80]:80s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:9.1f}')
PY
}
run narrow "rows dec" A=1
run wide "rows dec" HFAGP_DEV_PG_WIDE=1

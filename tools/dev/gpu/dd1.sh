cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
out=/tmp/tp; rm -rf $out; mkdir -p $out
HFAGP_DEV_DEC_DIRECT=$v rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python $GRAFT_REPO_ROOT/tools/dev/bench_train.py 2 8 3dmm tuned > $out/log.txt 2>&1
echo "dec_direct=$v"; tail -1 $out/log.txt | cut -c1-120
python - $out <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tiles_kernel" in r["Name"]:
            print(f'   {r["Name"][:80]:80s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:9.1f}')
PY
done

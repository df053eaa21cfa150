cd /tmp && export TMPDIR=/tmp
out=/tmp/qr; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python $GRAFT_REPO_ROOT/tools/dev/bench_train.py 2 8 3dmm > $out/log.txt 2>&1
python - $out <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "Cijk" in r["Name"] or "gram" in r["Name"] or "qr_" in r["Name"]:
            print(f'   {r["Name"][:100]:100s} calls/step={int(r["Calls"])/10:5.1f} avg_us={float(r["AverageNs"])/1e3:8.1f}')
PY

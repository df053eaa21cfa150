cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round5.py -x -q -k "adam or updated_weights" 2>&1 | grep -E "passed|failed|Error|assert|error" | head -5
cd /tmp && export TMPDIR=/tmp
for reg in "3dmm" "3dmm tuned"; do
out=/tmp/pw; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python $GRAFT_REPO_ROOT/tools/dev/bench_train.py 2 8 $reg > $out/log.txt 2>&1
python - $out <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "adam" in r["Name"]:
            print(f'   {r["Name"][:80]:80s} calls={r["Calls"]:>4s} avg_us={float(r["AverageNs"])/1e3:9.1f}')
PY
done

cd /tmp && export TMPDIR=/tmp
out=/tmp/tp; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python $GRAFT_REPO_ROOT/tools/dev/bench_train.py 2 8 3dmm tuned > $out/log.txt 2>&1
tail -1 $out/log.txt
python - $out <<'PY'
import csv, glob, sys
rows=[]
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((float(r["TotalDurationNs"])/1e6/10, int(r["Calls"])/10, float(r["AverageNs"])/1e3, r["Name"][:90]))
for ms, c, avg, n in sorted(rows, reverse=True)[:60]:
    print(f"{ms:7.3f} ms/step {c:6.1f} calls  avg {avg:8.1f} us  {n}")
print("total", sum(r[0] for r in rows))
PY

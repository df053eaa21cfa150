cd /tmp && export TMPDIR=/tmp
out=/tmp/enc; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python $GRAFT_REPO_ROOT/tools/dev/bench_train.py 2 8 rgb > $out/log.txt 2>&1
tail -1 $out/log.txt
python - $out <<'PY'
import csv, glob, sys
rows=[]
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((float(r["TotalDurationNs"])/1e6/10, int(r["Calls"])/10, r["Name"][:100]))
for ms, c, n in sorted(rows, reverse=True)[:200]:
    if any(k in n for k in ("weight_prep", "elementwise", "reduce_kernel", "Fill", "copy", "multi_tensor", "bias_act", "blur", "splitk")):
        print(f"{ms:7.3f} ms/step {c:6.1f} calls/step  {n}")
PY

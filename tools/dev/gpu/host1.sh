cd $GRAFT_REPO_ROOT
for reg in "3dmm" "rgb" "3dmm tuned" "rgb tuned"; do
  python tools/dev/bench_train.py 2 20 $reg 2>&1 | tail -1
done

cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "encoder or Encoder or rgb or batches_equal or dd_with" 2>&1 | grep -E "passed|failed|Error|assert|error" | head -12
for reg in "rgb" "rgb tuned"; do
  python tools/dev/bench_train.py 2 12 $reg 2>&1 | tail -1
done

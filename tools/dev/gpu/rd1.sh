cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_backward.py tests/test_gpu_round4.py tests/test_gpu_round3.py -x -q -k "backward or own_size or two_ranks or gradient or step" 2>&1 | grep -E "passed|failed|Error|assert|error" | head -12
for reg in "3dmm" "rgb"; do
  python tools/dev/bench_train.py 2 20 $reg 2>&1 | tail -1
done

cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py tests/test_gpu_backward.py tests/test_gpu_round2.py -x -q -k "tuned or parameter_gradients or wgrad or two_ranks or bucketed or own_size or trainer or step" 2>&1 | grep -E "passed|failed|Error|assert|error" | head -12
for reg in "3dmm tuned" "rgb tuned" "3dmm" "rgb"; do
  python tools/dev/bench_train.py 2 12 $reg 2>&1 | tail -1
done

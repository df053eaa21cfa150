cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py tests/test_gpu_round3.py -x -q -k "tuned or parameter_gradients or sort_gather or decoder" 2>&1 | grep -E "passed|failed|Error|assert|error" | head -8
bash tools/dev/gpu/dd1.sh 2>&1 | grep -v W2026 | head -3

cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round5.py -x -q -k "updated_weights or tuned_gen_update" 2>&1 | grep -E "passed|failed|Error|assert" | head -5
python tools/dev/gpu/check_tuned_cache.py 2>&1 | grep "^step"
python tools/dev/bench_train.py 2 12 3dmm tuned 2>&1 | tail -1
python tools/dev/bench_train.py 2 12 rgb tuned 2>&1 | tail -1

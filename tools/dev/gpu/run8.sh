cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round5.py -x -q -k "weight_prep_batch or updated_weights or tuned_gen_update" 2>&1 | grep -E "passed|failed|Error|assert|error" | head -8
python tools/dev/bench_train.py 2 12 3dmm tuned 2>&1 | tail -1
python tools/dev/bench_train.py 2 12 rgb tuned 2>&1 | tail -1

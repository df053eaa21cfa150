cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round5.py -x -q -k "adam or updated_weights" 2>&1 | grep -E "passed|failed|Error|assert|error" | head -8
python tools/dev/bench_train.py 2 12 3dmm tuned 2>&1 | tail -1
python tools/dev/bench_train.py 2 12 rgb tuned 2>&1 | tail -1
python tools/dev/bench_train.py 2 12 3dmm 2>&1 | tail -1
python tools/dev/bench_train.py 2 12 rgb 2>&1 | tail -1

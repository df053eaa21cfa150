cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_backward.py tests/test_gpu_round5.py tests/test_gpu_round4.py -x -q 2>&1 | grep -E "passed|failed|Error|error" | head -5
python tools/dev/bench_train.py 2 12 3dmm tuned 2>&1 | tail -1
python tools/dev/bench_train.py 2 12 3dmm 2>&1 | tail -1
python tools/dev/bench_train.py 2 12 rgb 2>&1 | tail -1
python tools/dev/bench_train.py 2 12 rgb tuned 2>&1 | tail -1
bash tests/micro/lanes48/repro.sh 200 2>&1 | tail -12

cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round5.py -x -q -k "channel_sum" 2>&1 | tail -3
python -m pytest tests/test_gpu_backward.py -x -q -k "weight_gradient or generator_parameter" 2>&1 | tail -3
python tools/dev/bench_train.py 2 10 3dmm tuned 2>&1 | tail -1
python tools/dev/bench_train.py 2 10 rgb tuned 2>&1 | tail -1

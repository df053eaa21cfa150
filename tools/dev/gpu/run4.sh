cd $GRAFT_REPO_ROOT
python tools/dev/bench_train.py 2 10 rgb tuned 2>&1 | tail -1
HFAGP_GRAD_INPLACE=0 python tools/dev/bench_train.py 2 10 rgb tuned 2>&1 | tail -1
python tools/dev/bench_train.py 2 30 rgb tuned 2>&1 | tail -1
HFAGP_GRAD_INPLACE=0 python tools/dev/bench_train.py 2 10 3dmm tuned 2>&1 | tail -1
python tools/dev/bench_train.py 2 30 3dmm tuned 2>&1 | tail -1

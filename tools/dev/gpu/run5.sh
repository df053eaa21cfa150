cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_backward.py -x -q -k "weight_gradient" 2>&1 | grep -E "passed|failed|Error|assert" | head
for sh in "2 256 256 128" "2 128 512 256" "2 64 512 512" "2 32 512 512"; do python tools/dev/bench_wgrad.py $sh up bf16x3 2>&1 | tail -1; done
python tools/dev/bench_train.py 2 10 3dmm tuned 2>&1 | tail -1

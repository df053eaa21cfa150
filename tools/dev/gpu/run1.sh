cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_backward.py -x -q -k "weight_gradient or generator_parameter" 2>&1 | tail -5
for sh in "2 256 256 256" "2 512 128 128" "2 128 256 256" "2 64 512 512" "2 32 512 512"; do python tools/dev/bench_wgrad.py $sh bf16x3 2>&1 | tail -1; done
python tools/dev/bench_wgrad.py 2 256 256 128 up 2>&1 | tail -1

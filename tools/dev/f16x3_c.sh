python -m pytest tests/test_gpu_parity.py -q -x -k "split_bf16 or f16 or torgb96" 2>&1 | tail -3
for prec in bf16x3 f16x3; do
python tests/bench_conv.py 8 256 256 256 1 0 300 $prec 2>&1 | grep conv
python tests/bench_conv.py 8 512 128 128 1 0 100 $prec 2>&1 | grep conv
python tests/bench_conv.py 8 256 256 128 2 0 200 $prec 2>&1 | grep conv
done

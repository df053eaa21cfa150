python -m pytest tests/test_gpu_parity.py -q -x -k "f16" 2>&1 | tail -15
for prec in bf16x3 f16; do
python tests/bench_conv.py 8 256 256 256 1 0 300 $prec
python tests/bench_conv.py 8 512 128 128 1 0 100 $prec
python tests/bench_conv.py 8 256 256 128 2 0 200 $prec
python tests/bench_conv.py 8 128 32 256 2 0 300 $prec
python tests/bench_conv.py 8 64 512 512 1 0 300 $prec
done

for w in 4 8; do for prec in bf16x3 f16; do
echo "== up_waves $w $prec"
HFAGP_DEV_UP_WAVES=$w python tests/bench_conv.py 8 256 256 128 2 0 200 $prec 2>&1 | grep conv
HFAGP_DEV_UP_WAVES=$w python tests/bench_conv.py 8 128 256 128 2 0 300 $prec 2>&1 | grep conv
HFAGP_DEV_UP_WAVES=$w python tests/bench_conv.py 8 64 512 256 2 0 300 $prec 2>&1 | grep conv
HFAGP_DEV_UP_WAVES=$w python tests/bench_conv.py 8 32 512 512 2 0 300 $prec 2>&1 | grep conv
HFAGP_DEV_UP_WAVES=$w python tests/bench_conv.py 8 128 32 256 2 0 300 $prec 2>&1 | grep conv
done; done
HFAGP_DEV_UP_WAVES=8 python -m pytest tests/test_gpu_parity.py -q -x -k "synthesis_layer_split or synthesis_layer_f16" 2>&1 | tail -3

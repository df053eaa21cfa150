for v in base noslp base noslp; do
  lib=hfa-gp_amd/libhfagp_hip.so; [ $v = noslp ] && lib=hfa-gp_amd/libhfagp_abl_noslp.so
  echo -n "$v: "; HFAGP_LIB_PATH=$PWD/$lib python bench.py --no-cpu-baseline --no-sweep --audio-frames 0 --no-fp32-leg --fit-frames 0 --fit3dmm-frames-per-rank 0 --no-lpips 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'fps', round(d['roofline_raymarch']['avg_launch_ms'],3), 'ms raymarch', d.get('train_step_ms'), d.get('train_step_ms_3dmm'))"
done

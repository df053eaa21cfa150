# ray-march kernel: timing at B=32 (default library vs an optional variant library) + the parity tests that exercise it
mkdir -p gpurun_out/rm
for v in base var base var; do
  lib=hfa-gp_amd/libhfagp_hip.so; [ $v = var ] && lib=${VARIANT_LIB:-hfa-gp_amd/libhfagp_hip.so}
  echo -n "$v: "; HFAGP_LIB_PATH=$PWD/$lib python tools/dev/bench_raymarch.py 32 10 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/rm/time.log
python -m pytest tests -m gpu -x -q -k "raymarch or render or synthesis or decoder" 2>&1 | tail -8 | tee gpurun_out/rm/tests.log

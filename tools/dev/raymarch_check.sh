# ray-march kernel: timing at B=32 (default vs variant: VARIANT_ENV="X=y" and/or VARIANT_LIB=<lib>) + the parity tests that exercise it
mkdir -p gpurun_out/rm
for v in base var base var; do
  lib=hfa-gp_amd/libhfagp_hip.so; envs=""
  if [ $v = var ]; then lib=${VARIANT_LIB:-hfa-gp_amd/libhfagp_hip.so}; envs="${VARIANT_ENV:-}"; fi
  echo -n "$v: "; env $envs HFAGP_LIB_PATH=$PWD/$lib python tools/dev/bench_raymarch.py ${RM_B:-32} 10 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/rm/time.log
[ -n "${SKIP_TESTS:-}" ] || python -m pytest tests -m gpu -x -q -k "raymarch or render or synthesis or decoder" 2>&1 | tail -8 | tee gpurun_out/rm/tests.log

for cfg in "2 256 256 256" "2 512 128 128" "2 128 512 512" "2 64 512 512" "2 256 128 128"; do
  for v in v1 v2; do
    if [ $v = v1 ]; then export HFAGP_DEV_WGRAD_V1=1; else unset HFAGP_DEV_WGRAD_V1; fi
    echo -n "$v: "; python tools/dev/bench_wgrad.py $cfg bf16x3 2>&1 | tail -1
  done
done
for cfg in "2 128 256 128" "2 256 256 128"; do
  for v in v1 v2; do
    if [ $v = v1 ]; then export HFAGP_DEV_WGRAD_V1=1; else unset HFAGP_DEV_WGRAD_V1; fi
    echo -n "$v: "; python tools/dev/bench_wgrad.py $cfg up 2>&1 | tail -1
  done
done

# Developer (GPU box): rows per thread of the stand-alone FIR epilogue kernel (8 shipped; 16 / 32 = fewer halo re-reads)
for v in 8 16 32 8 16 32; do
  lib=hfa-gp_amd/libhfagp_hip.so; [ $v != 8 ] && lib=hfa-gp_amd/libhfagp_abl_strip$v.so
  echo "== rows per thread $v"; HFAGP_FUSE_UP_FIR=0 HFAGP_LIB_PATH=$PWD/$lib UPFIR_LAYERS="256,256,128;128,32,256;64,512,256;16,512,512" python tools/dev/bench_upfir.py 32 f16x3 2>&1 | grep -- "->\|fuse_up_fir=0" | grep -v "sr_conv_precision=f16" | cut -c1-110
done

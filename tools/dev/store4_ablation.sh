#!/usr/bin/env bash
# Developer: what would 16-byte epilogue stores from a TRANSPOSED accumulator layout (lane = position, registers = channels)
# buy the 16-bit conv kernels?  Timing-only build (values land in the wrong places).  build HERE, `run` on the GPU box.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
csrc="${CSRC:-$here/hfa-gp_amd/csrc}"   # ablation variants: see tools/dev/patches/README.md
if [[ "${1:-build}" == "build" ]]; then
    bash "$csrc/build.sh" >/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DHFAGP_ABL_STORE4 -c "$csrc/modconv_bf16.hip" -o /tmp/mcb_store4.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DHFAGP_ABL_NOSTORE -c "$csrc/modconv_bf16.hip" -o /tmp/mcb_nostore.o
    for v in store4 nostore; do
        objs=(); for s in elementwise modconv upconv_fir torgb_skip raymarch backward raymarch_bwd wgrad wgrad_bf16 qr loss; do objs+=("$csrc/$s.o"); done
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" "/tmp/mcb_$v.o" -o "$here/hfa-gp_amd/libhfagp_abl_$v.so"
    done
else
    for v in base store4 nostore; do
        lib="$here/hfa-gp_amd/libhfagp_abl_$v.so"; [[ "$v" == base ]] && lib="$here/hfa-gp_amd/libhfagp_hip.so"
        for cfg in "32 256 256 256 1" "32 256 128 128 1" "32 512 128 128 1" "32 128 256 256 1" "32 256 256 128 2" "32 128 256 128 2" "32 64 512 256 2"; do
            echo -n "$v: "; HFAGP_LIB_PATH="$lib" python "$here/tools/dev/bench_conv.py" $cfg 0 20 f16x3 2>&1 | tail -1 | cut -c1-110
        done
    done
fi

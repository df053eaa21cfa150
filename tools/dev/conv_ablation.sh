#!/usr/bin/env bash
# Developer: ablation builds of the 16-bit conv kernel (what bounds it?).  Run HERE to build (cross-compile), the
# variants travel to the GPU box with gpurun; there:  bash tools/dev/conv_ablation.sh run
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
csrc="${CSRC:-$here/hfa-gp_amd/csrc}"   # ablation variants: see tools/dev/patches/README.md
variants=(base "nostore:-DHFAGP_ABL_NOSTORE" "old:-DHFAGP_LOADA_EARLY=0 -DHFAGP_B_EARLY=0" "aearly:-DHFAGP_LOADA_EARLY=1 -DHFAGP_B_EARLY=0" "bearly:-DHFAGP_LOADA_EARLY=0 -DHFAGP_B_EARLY=1"
          "nob:-DHFAGP_ABL_NOB" "noa:-DHFAGP_ABL_NOA" "nostage:-DHFAGP_ABL_NOSTAGE"
          "mfmaonly:-DHFAGP_ABL_NOB -DHFAGP_ABL_NOA -DHFAGP_ABL_NOSTAGE -DHFAGP_ABL_NOBAR" "presplit:-DHFAGP_ABL_PRESPLIT")
[[ -n "${ABL_VARIANTS:-}" ]] && read -r -a variants <<< "$ABL_VARIANTS"
if [[ "${1:-build}" == "build" ]]; then
    bash "$csrc/build.sh" >/dev/null
    for v in "${variants[@]}"; do
        name="${v%%:*}"; flags=""; [[ "$v" == *:* ]] && flags="${v#*:}"
        [[ "$name" == base ]] && continue
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $flags -c "$csrc/modconv_bf16.hip" -o "/tmp/mcb_$name.o" &
    done
    wait
    for v in "${variants[@]}"; do
        name="${v%%:*}"; [[ "$name" == base ]] && continue
        objs=(); for s in elementwise modconv torgb_skip raymarch backward raymarch_bwd wgrad wgrad_bf16 qr loss; do objs+=("$csrc/$s.o"); done
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" "/tmp/mcb_$name.o" -o "$here/hfa-gp_amd/libhfagp_abl_$name.so"
    done
    ls -la "$here"/hfa-gp_amd/libhfagp_abl_*.so
else
    for v in "${variants[@]}"; do
        name="${v%%:*}"
        lib="$here/hfa-gp_amd/libhfagp_abl_$name.so"; [[ "$name" == base ]] && lib="$here/hfa-gp_amd/libhfagp_hip.so"
        for prec in f16x3 f16; do
            echo -n "$name: "; HFAGP_LIB_PATH="$lib" python "$here/tools/dev/bench_conv.py" 8 256 256 256 1 0 300 $prec 2>&1 | tail -1
        done
        echo -n "$name: "; HFAGP_LIB_PATH="$lib" python "$here/tools/dev/bench_conv.py" 8 512 128 128 1 0 100 f16x3 2>&1 | tail -1
        echo -n "$name: "; HFAGP_LIB_PATH="$lib" python "$here/tools/dev/bench_conv.py" 8 64 512 512 1 0 300 f16x3 2>&1 | tail -1
        echo -n "$name: "; HFAGP_LIB_PATH="$lib" python "$here/tools/dev/bench_conv.py" 8 256 256 128 2 0 100 f16x3 2>&1 | tail -1
    done
fi

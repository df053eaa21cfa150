#!/usr/bin/env bash
# Developer: ablation builds of the fused up-sampling kernel (what does its epilogue cost?).  Run HERE to build
# (cross-compile), the variants travel to the GPU box with gpurun; there:  bash tools/dev/upfir_ablation.sh run
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
csrc="${CSRC:-$here/hfa-gp_amd/csrc}"   # ablation variants: see tools/dev/patches/README.md
variants=(base "nostore:1" "noexport:2" "nofir:4" "nostore_noexport:3" "nofir_nostore_noexport:7" "noldswrite:15" "noepilogue:16" "nonoise:32")
[[ -n "${ABL_VARIANTS:-}" ]] && read -r -a variants <<< "$ABL_VARIANTS"
if [[ "${1:-build}" == "build" ]]; then
    bash "$csrc/build.sh" >/dev/null
    for v in "${variants[@]}"; do
        name="${v%%:*}"; [[ "$name" == base ]] && continue
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DHFAGP_FIR_ABL="${v#*:}" ${ABL_EXTRA:-} -c "$csrc/upconv_fir.hip" -o "/tmp/uf_$name.o" &
    done
    wait
    for v in "${variants[@]}"; do
        name="${v%%:*}"; [[ "$name" == base ]] && continue
        objs=(); for s in elementwise modconv modconv_bf16 torgb_skip raymarch backward raymarch_bwd wgrad wgrad_bf16 qr loss; do objs+=("$csrc/$s.o"); done
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" "/tmp/uf_$name.o" -o "$here/hfa-gp_amd/libhfagp_abl_$name.so"
    done
    ls "$here"/hfa-gp_amd/libhfagp_abl_*.so
else
    for v in "${variants[@]}"; do
        name="${v%%:*}"
        lib="$here/hfa-gp_amd/libhfagp_abl_$name.so"; [[ "$name" == base ]] && lib="$here/hfa-gp_amd/libhfagp_hip.so"
        echo "== $name"; HFAGP_LIB_PATH="$lib" UPFIR_LAYERS="${UPFIR_LAYERS:-128,32,256;256,256,128}" UPFIR_NOSYN=1 python "$here/tools/dev/bench_upfir.py" ${UPFIR_B:-32} f16x3 2>&1 | grep -- "->"
    done
fi

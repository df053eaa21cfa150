#!/usr/bin/env bash
# Build libhfagp_hip.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${here}/../libhfagp_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function)
objs=()
for src in elementwise modconv modconv_bf16 upconv_fir torgb_skip raymarch backward raymarch_bwd wgrad wgrad_bf16 qr loss; do
    obj="${here}/${src}.o"
    extra=()
    # torgb_skip.hip: no SLP vectoriser (see the build note at the top of that file)
    [[ "$src" == torgb_skip ]] && extra=(-fno-slp-vectorize)
    if [[ ! -f "$obj" || "${here}/${src}.hip" -nt "$obj" || "${here}/common.h" -nt "$obj" || "${here}/modconv_plan.h" -nt "$obj" || "${here}/split_mfma.h" -nt "$obj" || "${here}/conv16_common.h" -nt "$obj" || "${here}/raymarch_common.h" -nt "$obj" || "${here}/build.sh" -nt "$obj" || "${here}/../../include/hfagp.h" -nt "$obj" ]]; then
        "$HIPCC" "${FLAGS[@]}" "${extra[@]}" ${HFAGP_EXTRA_FLAGS:-} -c "${here}/${src}.hip" -o "$obj" &
    fi
    objs+=("$obj")
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out"
echo "built $out"

#!/usr/bin/env bash
# Build libhfagp_hip.so for gfx950 (cross-compiles without a GPU).
# An object is rebuilt when the CONTENT of its source, of any header, of this script or of the flags changed (sha256 kept
# next to the object) — not by mtime, so a checkout / copy cannot leave a stale object behind.  HFAGP_CLEAN=1 rebuilds all.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${here}/../libhfagp_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -fno-slp-vectorize for EVERY unit (round 3): with the SLP vectoriser hipcc (ROCm 7.2) packs neighbouring fp32 arithmetic into
# v_pk_fma_f32 with swapped op_sel halves; in torgb_skip.hip that pattern sporadically dropped a result for lanes 48-63 on the
# MI355X (build note at the top of that file).  The root cause is not established, so the pattern is removed everywhere: measured
# cost none (render 800 / 799 vs 802 / 805 frames/s, ray march 6.55 ms either way, fitting steps equal: profiles/r03_no_slp.txt).
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize)
common_hash="$(cat "${here}"/*.h "${here}/../../include/hfagp.h" "${here}/build.sh" | sha256sum | cut -d' ' -f1)"
objs=()
built=0
for src in elementwise modconv modconv_bf16 smallconv upconv_fir torgb_skip raymarch backward raymarch_bwd wgrad wgrad_bf16 qr loss collective; do
    obj="${here}/${src}.o"
    extra=()
    want="$( (echo "${common_hash} ${FLAGS[*]} ${extra[*]:-} ${HFAGP_EXTRA_FLAGS:-}"; cat "${here}/${src}.hip") | sha256sum | cut -d' ' -f1)"
    have="$(cat "${obj}.sha256" 2>/dev/null || true)"
    if [[ "${HFAGP_CLEAN:-0}" == 1 || ! -f "$obj" || "$want" != "$have" ]]; then
        rm -f "${obj}.sha256"
        ( "$HIPCC" "${FLAGS[@]}" "${extra[@]}" ${HFAGP_EXTRA_FLAGS:-} -c "${here}/${src}.hip" -o "$obj" && echo "$want" > "${obj}.sha256" ) &
        built=$((built + 1))
    fi
    objs+=("$obj")
done
wait
for obj in "${objs[@]}"; do [[ -f "${obj}.sha256" ]] || { echo "build.sh: ${obj} failed to compile" >&2; exit 1; }; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out"
echo "built $out (${built} of ${#objs[@]} objects compiled)"

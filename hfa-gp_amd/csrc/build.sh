#!/usr/bin/env bash
# Build libhfagp_hip.so for gfx950 (cross-compiles without a GPU).
# An object is rebuilt when the CONTENT of its source, of any header, of this script or of the flags changed (sha256 kept
# next to the object) — not by mtime, so a checkout / copy cannot leave a stale object behind.  HFAGP_CLEAN=1 rebuilds all.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${here}/../libhfagp_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# NO PACKED FP32 ARITHMETIC in any unit: -fno-slp-vectorize (round 3) and -fno-vectorize (round 5).  hipcc (ROCm 7.2) packs
# neighbouring fp32 arithmetic into v_pk_fma / v_pk_mul / v_pk_add_f32; in torgb_skip.hip a v_pk_fma_f32 (op_sel) of the epilogue
# sporadically LOSES its low-half result for lanes 48-63 on the MI355X (one 16-lane beat: ~280 of 9.4 M outputs per call, always
# the same register of the unrolled epilogue, different tiles every run).  Round-5 reproducer (tests/micro/lanes48/, result in
# profiles/r05_lanes48_repro.txt) — the SAME source under six flag sets, 100-200 calls each, bit compare:
#   shipped flags 0 differences; SLP on: every call differs; SLP + MFMA latency fully padded with s_nops
#   (-amdgpu-mfma-padding-ratio=100): still every call -> NOT an MFMA write-back hazard (the accumulators are read hundreds of
#   instructions after the last MFMA anyway); SLP + every EXISTING s_waitcnt draining the load counters
#   (-amdgpu-waitcnt-load-forcezero): still failing, and a static model of vmcnt / lgkmcnt over the listing finds no uncovered use
#   (tools/dev/waitcnt_check.py) -> not a miscounted wait; SLP + an s_nop 0 before every instruction: unchanged, s_nop 3: ten times
#   rarer, a full s_waitcnt 0 before every instruction: gone -> the lost write needs a vector-memory return IN FLIGHT while the
#   packed op executes (lanes 48-63 are the last 16-lane beat of both): a register-file write conflict neither the ISA notes
#   available here nor LLVM's hazard recogniser list.  It cannot be fenced from source, so the instruction class is kept out of
#   the library: tests/test_kernel_resources.py counts v_pk_{fma,mul,add}_f32 in the ISA of EVERY unit (must be 0).  Cost: none
#   measured (render 800 / 799 vs 802 / 805 frames/s, fitting steps equal: profiles/r03_no_slp.txt; the loop vectoriser only touched
#   qr_refine_kernel and bias_act_bwd_kernel).
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize -fno-vectorize)
common_hash="$(cat "${here}"/*.h "${here}/../../include/hfagp.h" "${here}/build.sh" | sha256sum | cut -d' ' -f1)"
objs=()
built=0
for src in elementwise modconv modconv_bf16 smallconv upconv_fir upfir_lean torgb_skip raymarch backward raymarch_bwd raymarch_rows wgrad wgrad_bf16 qr loss collective; do
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

#!/usr/bin/env bash
# Run ON THE GPU BOX (via gpurun): reproducer for the sporadic "lanes 48-63 lose a result" bug of torgb_skip.hip.
# Rebuilds ONLY torgb_skip.o under several flag sets, links each with the shipped objects into /tmp/lanes48/<variant>.so and runs
# tests/micro/lanes48/run_variant.py against it (HFAGP_LIB_PATH).  What the variants decide:
#   noslp                 the shipped build (-fno-slp-vectorize): must show 0 differences
#   slp                   SLP vectoriser on (packed fp32 arithmetic in the epilogue): the failing build of round 2
#   slp_mfmapad           slp + -mllvm -amdgpu-mfma-padding-ratio=100 (s_nops fill the whole latency behind every MFMA):
#                         differences gone => an MFMA write-back hazard the recogniser misses
#   slp_waitzero          slp + -mllvm -amdgpu-waitcnt-forcezero (every s_waitcnt waits for everything): differences gone => a
#                         memory return (scratch / global / LDS) consumed too early — or merely that serialising the stream hides a
#                         timing hazard; the next three tell those apart
#   slp_loadzero          slp + -mllvm -amdgpu-waitcnt-load-forcezero (the EXISTING waits drain the load counters completely)
#   slp_snop0 / slp_snop3 slp + -mllvm -amdgpu-snop-padding=1 / 4 (an s_nop in front of every instruction: issue spacing only, no
#                         memory wait added): differences gone => an issue-timing (VALU read-after / write-after) hazard
# (SLP turns up2_taps' parity swap of (y0, y1) / (wy0, wy1) into a DYNAMICALLY indexed 2-vector, which the back end lowers to
#  scratch_store at a computed offset + scratch_load of both elements with no wait in between: the ISA count of scratch
#  instructions is printed per variant.)
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}"
src="$R/hfa-gp_amd/csrc"; out=/tmp/lanes48; mkdir -p "$out" "$R/gpurun_out"
runs="${1:-300}"
base=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function)
declare -A flags=(
  [noslp]="-fno-slp-vectorize"
  [slp]=""
  [slp_mfmapad]="-mllvm -amdgpu-mfma-padding-ratio=100"
  [slp_waitzero]="-mllvm -amdgpu-waitcnt-forcezero"
  [slp_loadzero]="-mllvm -amdgpu-waitcnt-load-forcezero"
  [slp_snop0]="-mllvm -amdgpu-snop-padding=1"
  [slp_snop3]="-mllvm -amdgpu-snop-padding=4"
)
objs=(); for u in elementwise modconv modconv_bf16 smallconv upconv_fir raymarch backward raymarch_bwd wgrad wgrad_bf16 qr loss collective; do objs+=("$src/$u.o"); done
log="$R/gpurun_out/lanes48_repro.txt"; : > "$log"
for v in ${VARIANTS:-noslp slp slp_mfmapad slp_waitzero slp_loadzero slp_snop0 slp_snop3}; do
  /opt/rocm/bin/hipcc "${base[@]}" ${flags[$v]} -c "$src/torgb_skip.hip" -o "$out/torgb_$v.o" 2>/dev/null || { echo "variant $v: compile failed" | tee -a "$log"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "${objs[@]}" "$out/torgb_$v.o" -o "$out/$v.so" || { echo "variant $v: link failed" | tee -a "$log"; continue; }
  npk=$(/opt/rocm/bin/hipcc "${base[@]}" ${flags[$v]} -S --cuda-device-only "$src/torgb_skip.hip" -o - 2>/dev/null | grep -c "v_pk_fma_f32\|v_pk_mul_f32\|v_pk_add_f32")
  nscr=$(/opt/rocm/bin/hipcc "${base[@]}" ${flags[$v]} -S --cuda-device-only "$src/torgb_skip.hip" -o - 2>/dev/null | grep -c "scratch_")
  echo "variant $v: flags '${flags[$v]}': $npk packed fp32 instructions, $nscr scratch instructions in the ISA" | tee -a "$log"
  HFAGP_VARIANT=$v HFAGP_LIB_PATH="$out/$v.so" python "$R/tests/micro/lanes48/run_variant.py" "$runs" 2>&1 | grep "^variant" | tee -a "$log"
done

#!/usr/bin/env bash
# Host-side sanitizer build: libhfagp with -fsanitize=address,undefined on the HOST code (device code
# unchanged: -fno-gpu-sanitize), then tools/sanitize/host_driver.c under ASan + UBSan.  Needs no GPU.
# usage: tools/sanitize/run.sh [build dir]
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
csrc="$here/hfa-gp_amd/csrc"
out="${1:-$(mktemp -d /tmp/hfagp_asan.XXXXXX)}"
mkdir -p "$out"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
CLANG="${CLANG:-/opt/rocm/lib/llvm/bin/clang}"
SAN=(-fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -fno-sanitize-recover=undefined)
objs=()
for src in "$csrc"/*.hip; do
    o="$out/$(basename "${src%.hip}").o"
    "$HIPCC" --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Wno-unused-function "${SAN[@]}" -c "$src" -o "$o" &
    objs+=("$o")
done
wait
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${SAN[@]}" "${objs[@]}" -o "$out/libhfagp_asan.so"
"$CLANG" -std=c99 -g "${SAN[@]/-fno-gpu-sanitize/}" -I"$here/include" "$here/tools/sanitize/host_driver.c" -L"$out" -lhfagp_asan \
    -Wl,-rpath,"$out" -Wl,-rpath,/opt/rocm/lib -o "$out/host_driver"
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 "$out/host_driver"

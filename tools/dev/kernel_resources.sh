#!/usr/bin/env bash
# Developer: per-kernel register / spill / occupancy table of one csrc unit (cross-compiles, no GPU).  usage: kernel_resources.sh wgrad_bf16 [filter]
here="$(cd "$(dirname "$0")/../../hfa-gp_amd/csrc" && pwd)"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize ${HFAGP_EXTRA_FLAGS:-} -c "$here/$1.hip" -o /tmp/kr_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void hfagp::", "")}
        rows.append(cur)
    for key, pat in (("vgpr", r"VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sgpr", r" SGPRs: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = m.group(1)
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for r in rows:
    if flt in r["name"]:
        print("%-64s vgpr %4s agpr %4s spill %4s scratch %5s occ %s" % (r["name"][:64], r.get("vgpr"), r.get("agpr"), r.get("spill"), r.get("scratch"), r.get("occ")))
' "${2:-}"
rm -f /tmp/kr_$$.o

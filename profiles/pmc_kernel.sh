#!/usr/bin/env bash
# Run ON THE GPU BOX: PMC passes (own runs, no other tracing) for one script; prints per-kernel averages.
# usage: bash profiles/pmc_kernel.sh <kernel-substring> <script> [args...]
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
kern="$1"; shift
out="$R/gpurun_out/pmc_$$"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
passes=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA"
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAIT_INST_LDS"
 "SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT"
 "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE TA_TA_BUSY_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
)
i=0
for p in "${passes[@]}"; do
  rocprofv3 --pmc $p --kernel-trace --output-format csv -d "$out/p$i" -o x -- python "$R/$1" "${@:2}" > "$out/p$i.log" 2>&1
  i=$((i+1))
done
python - "$out" "$kern" <<'PY'
import csv, glob, sys
from collections import defaultdict
out, kern = sys.argv[1], sys.argv[2]
agg = defaultdict(lambda: [0, 0.0])
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k in sorted(agg):
    n, v = agg[k]
    print(f"{k:32s} launches={n:3d} avg={v/n:.4g}")
PY

#!/usr/bin/env bash
# Run ON THE GPU BOX (via gpurun):  bash profiles/run_profile.sh <tag> [bench args]
# 1) kernel trace + stats of the bench command, 2)-4) PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy) in their own runs.
set -uo pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
tag="${1:-r01}"; shift || true
args=("$@"); [[ ${#args[@]} -eq 0 ]] && args=(--steps 10 --warmup 3 --no-cpu-baseline)
# raw rocprofv3 output stays on the box (/tmp): gpurun copies back at most 64 MiB, only the summaries travel
keep="$R/gpurun_out/prof_$tag"
out="/tmp/prof_$tag"
rm -rf "$out"; mkdir -p "$out" "$keep"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- python "$R/bench.py" "${args[@]}" --detail "$out/bench_detail_profiled.json" > "$out/trace.log" 2>&1
grep '^{' "$out/trace.log" > "$out/bench_line_profiled.json" || true
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -o bench -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-sweep --audio-frames 0 > "$out/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -o bench -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-sweep --audio-frames 0 > "$out/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/pmc_mfma" -o bench -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-train --no-sweep --audio-frames 0 > "$out/pmc_mfma.log" 2>&1
python "$R/bench.py" "${args[@]}" --detail "$out/bench_detail_unprofiled.json" > "$out/bench_unprofiled.log" 2>&1
find "$out" -name "*.csv" | head -20
python "$R/profiles/summarize.py" "$out" "$tag" > "$out/summary_$tag.md" 2>&1 || true
cp "$out/summary_$tag.md" "$out/traffic.json" "$out/bench_unprofiled.log" "$out/bench_line_profiled.json" "$out/bench_detail_unprofiled.json" "$keep/" 2>/dev/null
cp "$(find "$out/trace" -name "*kernel_stats.csv" | head -1)" "$keep/kernel_stats.csv" 2>/dev/null
tail -40 "$out/summary_$tag.md"

"""Condense the rocprofv3 CSVs written by run_profile.sh into a small markdown summary."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(root, pat):
    hits = sorted(glob.glob(os.path.join(root, "**", pat), recursive=True))
    return hits[0] if hits else None


def short(name):
    name = name.replace("hfagp::", "").replace("void ", "")
    return name.split("(")[0][:60]


def main():
    out, tag = sys.argv[1], sys.argv[2]
    print(f"# rocprofv3 summary {tag}\n")
    js = os.path.join(out, "bench_unprofiled.log")
    detail = os.path.join(out, "bench_detail_unprofiled.json")      # (round 6: the printed line is compact, the tables live here)
    if os.path.exists(js):
        for line in open(js):
            if line.startswith("{"):
                d = json.loads(line)
                if os.path.exists(detail):
                    d = json.load(open(detail))
                r32 = d.get("roofline_fp32_exact")
                extra = (f"; exact-fp32 leg {d['value_fp32_exact']:.1f} frames/s, modconv_kernel {r32['achieved']:.1f} "
                         f"TFLOP/s ({r32['frac']:.3f} of 157.3)") if r32 else ""
                print(f"bench (un-profiled): {d['value']:.1f} {d['unit']}, {d['ms_per_step']:.2f} ms/step, "
                      f"B={d['config']['frames_per_step_per_gpu']}, conv precision {d['config'].get('conv_precision')}; "
                      f"{d['roofline']['kernel']} {d['roofline']['achieved']:.1f} TFLOP/s "
                      f"({d['roofline']['frac']:.3f} of {d['roofline']['peak']:.0f}){extra}; raymarch "
                      f"{d['roofline_raymarch']['avg_launch_ms']:.3f} ms/launch = {d['roofline_raymarch']['frac']:.3f} of its "
                      f"L2-gather + decoder-MFMA floor (gather rate {d['roofline_raymarch']['gather_rate_GBps_survey8d']:.0f} GB/s)\n")
    stats = find(os.path.join(out, "trace"), "*kernel_stats.csv")
    if stats:
        print("## kernel stats (rocprofv3 --kernel-trace --stats), bench command\n")
        print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
        for r in csv.DictReader(open(stats)):
            print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
                  f"{float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
    traffic = {}
    for kind, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        f = find(os.path.join(out, kind), "*counter_collection.csv")
        if not f:
            continue
        agg = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        print(f"\n## {counter} per launch (KiB as reported; gfx950: FETCH_SIZE under-reports wide reads 2x, see MI355X_MICROARCH.md)\n")
        print("| kernel | launches | avg per launch (MB, raw) |\n|---|---|---|")
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"| {k} | {n} | {v / n * 1024 / 1e6:.2f} |")
            traffic.setdefault(k, {})[counter] = v / n * 1024          # bytes per launch, as reported (KiB units)
    # matrix-pipe utilisation per kernel: SQ_VALU_MFMA_BUSY_CYCLES counts busy cycles summed over the 1024 SIMDs
    # (= 32 x N_mfma for a 32x32x16 16-bit MFMA); GRBM_GUI_ACTIVE is summed over the 8 XCDs
    f = find(os.path.join(out, "pmc_mfma"), "*counter_collection.csv")
    if f:
        agg = defaultdict(lambda: defaultdict(float))
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
        print("\n## MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), summed over the kernel's launches\n")
        print("| kernel | MFMA instructions | MFMA busy |\n|---|---|---|")
        for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
            busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
            if busy > 0 and act > 0:
                print(f"| {k} | {c.get('SQ_INSTS_MFMA', 0.0):.3g} | {busy / (128.0 * act):.3f} |")
    # HBM/fabric bytes per launch for bench.py's `traffic` field: FETCH_SIZE doubled (gfx950 reports half of a wide
    # coalesced read, MI355X_MICROARCH.md section HBM) + WRITE_SIZE
    batch = 8
    try:
        batch = json.loads([l for l in open(js) if l.startswith("{")][0])["config"]["frames_per_step_per_gpu"]
    except Exception:
        pass
    out_t = {"tag": tag, "batch": batch, "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of bench.py "
             "--no-train; bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes); averages over all launches of "
             "the kernel in one synthesis"}
    for k, v in traffic.items():
        if k.startswith(("modconv_kernel<2, 2, 2, 2>", "modconv_bf16_kernel<2, 2, 9", "modconv_bf16_kernel<4, 2, 9", "upconv_bf16_kernel",
                         "upfir_lean_kernel", "upfir_epilogue_kernel", "torgb_skip_kernel", "raymarch_kernel")):
            out_t[k] = {"fetch_raw": v.get("FETCH_SIZE"), "write": v.get("WRITE_SIZE"),
                        "hbm_bytes": 2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)}
    json.dump(out_t, open(os.path.join(out, "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

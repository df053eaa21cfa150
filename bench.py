#!/usr/bin/env python
"""bench.py — rendered 512^2 frames/s at 96 depth samples (BASELINE.json metric, config 2).

One "step" = one pass of the hot path over one batch of synthetic input:
``generator.synthesis(ws[B,14,512], c[B,25], noise_mode='const')`` for the
``ffhq512_128`` preset (random-init EG3D weights, random latents + gaussian cameras,
explicit sampling uniforms), forward only, inputs resident in HBM before the
timed region.  N>1: one process per GPU, frames are independent → weak scaling,
no data-path collective; only the timing is reduced (MAX).

``python bench.py --gpus N`` WITHOUT a launcher (WORLD_SIZE unset) re-executes itself under
``torch.distributed.run`` with N ranks (the reference spawns one process per GPU itself:
/root/reference/code/train_rgb.py:196-202, ``ddp_setup`` :53-57); under a launcher it checks that the
process group really has N ranks.  ``n_gpus`` on the line is what the process group reports.

Secondary legs on the same JSON line (none of them is ``value``): other conv precisions, batch sweep,
the fitting step of BASELINE configs 3 (RGB-driven, Encoder in the step) and 4 (3DMM-driven) with the
all-reduce of the shared gradients, a 500-frame synthetic fit (config 3 as written), the batched
audio-driven reenactment of config 5, the CPU oracle baseline.

Prints ONE JSON line on rank 0 (see the contract in the task description).
"""
from __future__ import annotations

import argparse
import gc
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
L2_PEAK_GBS = 34500.0        # MI355X_MICROARCH.md: L2 aggregate ~34.5 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32 / 16x16x4_f32 dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_bf16 dense peak
SPLIT_MFMAS = {"f16x3": 3, "bf16x3": 3, "bf16x6": 6}   # 16-bit MFMAs per algorithmic (fp32) product on the split paths
SPLIT_ELEM = {"f16x3": "f16", "bf16x3": "bf16", "bf16x6": "bf16"}
MFMA_F16_PEAK_TFLOPS = 2500.0   # v_mfma_f32_32x32x16_f16 dense peak (same rate as bf16)


LINE_BYTE_BUDGET = 6000      # the driver parses the ONE stdout line out of a bounded tail (round 5's 21.6 KB line came back unparsed)


def _sig(v, digits=5):
    """floats rounded to `digits` significant digits (the line is a report, not an archive: bench_detail.json keeps every bit)."""
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float(f"{v:.{digits}g}")
    if isinstance(v, dict):
        return {k: _sig(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_sig(x, digits) for x in v]
    return v


def _rows_stats() -> dict:
    """Scratch bytes / frame chunks / path (sort + gather rows or scatter fallback) of the LAST ray-march backward call of this
    process (hfa_gp_amd.ops.ROWS_STATS)."""
    from hfa_gp_amd import ops
    return ops.ROWS_STATS


def compact_line(full: dict) -> dict:
    """The ONE line rank 0 prints: the contract's keys, the dominant kernel's `roofline`, `cpu_baseline`, and a few dozen scalars
    (fitting steps, family roofline fractions, the other legs' headline numbers).  Everything else `main` measured — per-layer
    and per-kernel tables, percentiles, phases, prose — goes to bench_detail.json (and stderr).  tests/test_bench_line.py holds the
    result under LINE_BYTE_BUDGET bytes on canned leg results."""
    def get(d, *path, default=None):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return default
            d = d[k]
        return d

    cfg = full.get("config", {})
    prec = cfg.get("conv_precision", "f16x3")
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline")}
    line["dtype"] = "f32" if prec == "fp32" else f"f32 (conv GEMM products as {prec}: split 16-bit operands, fp32 accumulate)"
    line["data"] = full.get("data", "synthetic")
    line["config"] = {"workload": f"{str(cfg.get('workload', '')).split(':')[0]}: synthesis fwd, 512^2 out, 128^2 rays x (48+48) samples, "
                                  f"random-init weights (BASELINE config 2)",
                      "frames_per_step_per_gpu": cfg.get("frames_per_step_per_gpu"), "parallelism": cfg.get("parallelism"),
                      "conv_precision": prec}
    roof = full.get("roofline", {})
    line["roofline"] = {k: roof.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms",
                                                 "launches")}
    cb = full.get("cpu_baseline")
    if cb is not None:
        line["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "cpu_model", "sample") if k in cb}
        for k in ("one_frame_latency", "n1"):
            if k in cb:
                line["cpu_baseline"][k] = {q: cb[k].get(q) for q in ("value", "cores", "s_per_frame") if q in cb[k]}
        if "parity_vs_oracle" in cb:
            line["cpu_baseline"]["parity_vs_oracle"] = {q: cb["parity_vs_oracle"].get(q) for q in ("max_abs", "mse")}
    if len(full.get("per_rank_frames_per_s") or []) > 1:
        line["per_rank_frames_per_s"] = full["per_rank_frames_per_s"]
        line["per_rank_spread"] = full.get("per_rank_spread")
    sc = {"step_ms_median": get(full, "step_ms", "median"),
          "raymarch_frac_of_floor": get(full, "roofline_raymarch", "frac"),
          "raymarch_ms_per_launch": get(full, "roofline_raymarch", "avg_launch_ms"),
          "raymarch_hbm_frac_of_peak": get(full, "roofline_raymarch", "hbm_counter_frac_of_peak"),
          "all_conv_gemms_frac": get(roof, "all_conv_gemms", "frac"),
          "up_conv_frac": get(roof, "up_conv", "frac")}
    ups = roof.get("up_layers") or []
    if ups:
        sc["up_layers_ms_per_step"] = sum(u["ms_per_launch"] for u in ups)
        for u in ups:
            if u["layer"] in ("256->128@512", "32->256@256:fused", "32->256@256"):
                sc["up_layer_ms[" + u["layer"] + "]"] = u["ms_per_launch"]
    for b, row in (full.get("batch_sweep") or {}).items():
        if b in ("1", "8", "128") and "frames_per_s_per_gpu" in row:
            sc[f"frames_per_s_b{b}"] = row["frames_per_s_per_gpu"]
    for k in ("value_f16_sr", "value_f16", "value_f16_sr_f16_storage", "value_bf16x3", "value_fp32_exact"):
        if k in full:
            sc[k] = full[k]
    sc["value_f16x2"] = get(full, "tf32_class_leg", "value")
    sc["f16_sr_frac"] = get(full, "roofline_f16_sr", "frac")
    for k in ("train_step_ms", "train_step_ms_3dmm", "train_step_ms_generator_tuned", "train_step_ms_3dmm_generator_tuned"):
        sc[k] = full.get(k)
    sc["train_step_ms_lpips"] = get(full, "train_step_ms_lpips", "step_ms")
    sc["train_step_ms_3dmm_b1"] = get(full, "train_step_ms_3dmm_by_batch", "1")
    sc["train_frames_per_step"] = get(full, "train_config", "frames_per_step_per_gpu")
    sc["fit_rgb_ms_per_step"] = get(full, "fit_rgb", "ms_per_step")
    sc["fit_rgb_frames_per_s"] = get(full, "fit_rgb", "frames_per_s")
    sc["fit_3dmm_sharded_ms_per_step"] = get(full, "fit_3dmm_sharded", "ms_per_step")
    sc["fit_3dmm_sharded_frames_per_s"] = get(full, "fit_3dmm_sharded", "frames_per_s")
    sc["audio_reenactment_frames_per_s"] = get(full, "audio_reenactment", "frames_per_s")
    sc["allreduce_us_3dmm"] = get(full, "allreduce_us", "3dmm")
    sc["allreduce_us_rgb_tuned"] = get(full, "allreduce_us", "rgb_generator_tuned")
    line.update({k: v for k, v in sc.items() if v is not None})
    # the backward kernel families: fraction of their roofline + ms per step, frozen (3DMM) and generator-tuned (3DMM, RGB)
    rt = full.get("roofline_train") or {}
    fam = {}
    for tag, node in (("3dmm", rt.get("3dmm")), ("rgb", rt.get("rgb")), ("tuned_3dmm", get(rt, "tuned", "3dmm")),
                      ("tuned_rgb", get(rt, "tuned", "rgb"))):
        if not node:
            continue
        row = {}
        for key in ("bwd_data_gemms", "wgrad_gemms", "wgrad", "wgrad_up", "wgrad_1x1", "pointwise_bwd", "raymarch_bwd"):
            if key in node:
                row[key] = [node[key].get("frac"), node[key].get("ms_per_step")]
        fam[tag] = row
    if fam:
        line["roofline_train_frac_ms"] = fam
    line["leg_seconds"] = full.get("leg_seconds")
    line["detail"] = "bench_detail.json (every table of this run; also on stderr)"
    line = _sig(line)
    # never lose the line to its own length: drop the least important groups until it fits
    for victim in ("leg_seconds", "roofline_train_frac_ms", "per_rank_frames_per_s"):
        if len(json.dumps(line)) <= LINE_BYTE_BUDGET:
            break
        line.pop(victim, None)
    return line


def write_detail(full: dict, where=None) -> list:
    """bench_detail.json next to the script (and under gpurun_out/ so a gpurun call brings it back) or the file `where` names; the
    full object also goes to stderr.  Returns the paths written."""
    paths = []
    for path in ([where] if where else [os.path.join(ROOT, "bench_detail.json"), os.path.join(ROOT, "gpurun_out", "bench_detail.json")]):
        try:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            with open(path, "w") as f:
                json.dump(full, f, indent=1)
            paths.append(path)
        except OSError:
            pass
    sys.stderr.write("[bench detail] " + json.dumps(full) + "\n")
    sys.stderr.flush()
    return paths


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32,
                    help="frames per step per GPU (throughput: 8 -> 16 -> 32 frames per synthesis call render "
                         "668 -> 706 -> 728 frames/s; the sweep leg lists 1 ... 32)")
    ap.add_argument("--preset", default="ffhq512_128")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the fitting-step legs (train_step_ms ...)")
    ap.add_argument("--train-batch", type=int, default=2, help="frames per fitting step per GPU")
    ap.add_argument("--train-steps", type=int, default=10)
    ap.add_argument("--fit-frames", type=int, default=500,
                    help="frames of the synthetic RGB-driven fit (BASELINE config 3; 0 = skip): one pass over them")
    ap.add_argument("--fit3dmm-frames-per-rank", type=int, default=250,
                    help="frames per rank of the frame-sharded 3DMM-driven fit (BASELINE config 4: 2000 frames on 8 GPUs; "
                         "0 = skip)")
    ap.add_argument("--audio-frames", type=int, default=256,
                    help="frames of the batched audio-driven reenactment leg (BASELINE config 5; 0 = skip)")
    ap.add_argument("--no-lpips", dest="lpips", action="store_false",
                    help="skip the fitting-step leg with the LPIPS(alex) term (seeded random weights: the COST of the reference "
                         "objective l2 + lpips, trainer_rgb.py:86-91 — on by default)")
    ap.add_argument("--dist-timeout", type=int, default=300,
                    help="seconds a collective may wait for a peer before the job fails with a message (N > 1)")
    ap.add_argument("--no-sweep", action="store_true",
                    help="skip the batch-size sweeps (render B = 1, 4, 16; fitting step B = 1, 4; SURVEY.md section 8d)")
    ap.add_argument("--cpu-runs", type=int, default=3, help="timed one-frame oracle runs (BASELINE.md section 3 says 5; 3 keep the leg short)")
    ap.add_argument("--cpu-warmup", type=int, default=1, help="oracle warm-up runs (BASELINE.md section 3 says 2)")
    ap.add_argument("--no-cpu-n1", dest="cpu_n1", action="store_false",
                    help="skip the ONE-thread run of the oracle (BASELINE.md section 3: n = 1; ~5 s)")
    ap.add_argument("--precision", default=None, choices=["fp32", "f16x3", "bf16x3", "bf16x6"],
                    help="conv GEMM arithmetic of the headline leg (default: the preset's conv_precision)")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the second render leg on the exact fp32 kernel")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend (nccl = RCCL; gloo + --share-device: developer check of the N > 1 code "
                         "path on a 1-GPU box, numbers not comparable)")
    ap.add_argument("--share-device", action="store_true", help="developer: every rank uses cuda:0")
    ap.add_argument("--detail", default=None,
                    help="file for the full result object (default: bench_detail.json next to this script and under gpurun_out/)")
    ap.add_argument("--no-f16-leg", action="store_true",
                    help="skip the legs with the super-resolution / all convs on the single-pass fp16 MFMA path")
    return ap.parse_args()


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(n: int, argv, port: int):
    """The command line `--gpus N` re-executes when no launcher set WORLD_SIZE (one rank per GPU, rendezvous on
    127.0.0.1) — the driver's own form of the launch."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + list(argv)


def rccl_choices(log_path, n_ranks):
    """{payload bytes: {"algo", "proto", "time_us"}} from RCCL's TUNING lines of this rank's NCCL_DEBUG_FILE
    ("AllReduce: N Bytes -> Algo a proto p time t"; enum names of nccl.h since 2.19), or a note saying why there are none —
    with ONE rank RCCL copies in place and never runs its tuning model."""
    import re
    algos = {0: "Tree", 1: "Ring", 2: "CollNetDirect", 3: "CollNetChain", 4: "NVLS", 5: "NVLSTree"}
    protos = {0: "LL", 1: "LL128", 2: "Simple"}
    if not log_path or not os.path.exists(log_path):
        return {"note": f"no RCCL log ({log_path}): NCCL_DEBUG / NCCL_DEBUG_FILE were set by the caller, or the backend is not nccl"}
    found, version = {}, None
    pat = re.compile(r"AllReduce: (\d+) Bytes -> Algo (\d+) proto (\d+) time ([0-9.eE+-]+)")
    with open(log_path, errors="replace") as f:
        for line in f:
            m = pat.search(line)
            if m:
                found[m.group(1)] = {"algo": algos.get(int(m.group(2)), m.group(2)), "proto": protos.get(int(m.group(3)), m.group(3)),
                                     "model_time_us": float(m.group(4))}
            elif "RCCL version" in line or "NCCL version" in line:
                version = line.strip().split("INFO")[-1].strip()
    if found:
        return {"by_payload_bytes": found, "rccl": version, "log": log_path}
    return {"note": ("one rank: RCCL short-circuits the collective (in-place copy), its tuning model never runs" if n_ranks == 1 else
                     "no 'Bytes -> Algo' line in the log (RCCL built without the TUNING trace?)"), "rccl": version, "log": log_path}


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_topology() -> dict:
    """sockets / physical cores / logical CPUs of this host from /proc/cpuinfo (the GPU box: 2 x 64 cores, 256 threads)."""
    phys, cores_per, logical = set(), 0, 0
    try:
        pid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                logical += 1
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
                phys.add(pid)
            elif line.startswith("cpu cores"):
                cores_per = int(line.split(":", 1)[1])
    except (OSError, ValueError):
        pass
    return {"sockets": len(phys) or 1, "cores_per_socket": cores_per, "logical_cpus": logical}


def _cpu_parallel_worker(rank, threads, frames, preset, state, barrier, out):
    """One of P oracle processes of `cpu_baseline`'s `parallel` leg: `threads` intra-op threads, one warm-up frame, then
    `frames` B = 1 synthesis calls between a common barrier and its own finish time."""
    import torch
    torch.set_num_threads(threads)
    from oracle import eg3d_oracle as O
    from hfa_gp_amd.config import PRESETS
    from hfa_gp_amd.synthetic import make_inputs
    cfg = PRESETS[preset]()
    ws, c, us, ui = make_inputs(cfg, 1, seed=10 + rank)
    with torch.no_grad():
        O.synthesis(state, cfg, ws, c, us, ui)
        barrier.wait()
        t0 = time.time()
        for _ in range(frames):
            img = O.synthesis(state, cfg, ws, c, us, ui)["image"]
        out.put((rank, t0, time.time(), bool(torch.isfinite(img).all())))


def cpu_parallel(cfg, state, phys: int, frames: int = 2, threads: int = 16):
    """the ATen CPU ops of this path do not scale with threads at batch 1 (one frame on 128 threads is no
    faster than on one), so "128 cores" above is a thread count.  This leg keeps all physical cores BUSY instead: P = cores /
    16 oracle processes of 16 threads each, every one rendering `frames` frames of its own; frames/s = P * frames / wall."""
    import torch.multiprocessing as mp
    threads = max(1, min(threads, phys))
    procs = max(1, phys // threads)
    ctx = mp.get_context("spawn")
    barrier, out = ctx.Barrier(procs), ctx.Queue()
    shared = {k: v.share_memory_() for k, v in state.items()}
    ps = [ctx.Process(target=_cpu_parallel_worker, args=(r, threads, frames, cfg.name, shared, barrier, out)) for r in range(procs)]
    for p_ in ps:
        p_.start()
    import queue
    rows, deadline = [], time.time() + 600
    while len(rows) < procs:
        try:
            rows.append(out.get(timeout=2))
        except queue.Empty:
            dead = [p_.exitcode for p_ in ps if p_.exitcode not in (None, 0)]
            if dead or time.time() > deadline:          # a worker that died (or a wedged one) must not hold the run for minutes
                for p_ in ps:
                    if p_.is_alive():
                        p_.terminate()
                raise RuntimeError(f"oracle worker exit codes {dead}" if dead else "oracle workers timed out")
    for p_ in ps:
        p_.join(timeout=60)
    t0, t1 = min(r[1] for r in rows), max(r[2] for r in rows)
    assert all(r[3] for r in rows)
    return {"value": procs * frames / (t1 - t0), "unit": "frames/s", "processes": procs, "threads_per_process": threads,
            "cores": procs * threads, "frames_per_process": frames, "wall_s": t1 - t0,
            "s_per_frame_per_process": (t1 - t0) / frames,
            "what": "all physical cores busy: independent oracle processes, one frame at a time each (frame-parallel, as the GPU "
                    "path scales); the B = 1 figures above are the latency of ONE frame on the whole box"}


def cpu_baseline(cfg, state, runs: int, warmup: int, n1: bool, check=None):
    """Oracle (kind 'port') timed on this box's host cores: B=1 synthesis of the same workload; median of `runs`
    after `warmup`; stage split backbone / ray-march / super-resolution.
    `check(ws, c, us, ui)` -> the HIP path's image of the same inputs (CPU tensor): the oracle's last image is then also the
    CHECKER of this very run (`parity_vs_oracle`: max abs / MSE of the 512^2 image) — never the thing measured."""
    import torch
    from oracle import eg3d_oracle as O
    from hfa_gp_amd.synthetic import make_inputs
    ws, c, us, ui = make_inputs(cfg, 1, seed=10)
    res = cfg.neural_rendering_resolution
    last = {}

    def one():
        t = [time.perf_counter()]
        planes = O.backbone_synthesis(state, cfg, ws)
        t.append(time.perf_counter())
        o, d = O.ray_sampler(c[:, :16].reshape(-1, 4, 4), c[:, 16:25].reshape(-1, 3, 3), res)
        p5 = planes.reshape(1, 3, cfg.plane_channels, planes.shape[-2], planes.shape[-1])
        feat, _, _ = O.importance_renderer(state, cfg, p5, o, d, us, ui)
        t.append(time.perf_counter())
        fi = feat.permute(0, 2, 1).reshape(1, feat.shape[-1], res, res).contiguous()
        last["image"] = O.superresolution(state, cfg, fi[:, :3], fi, ws)
        t.append(time.perf_counter())
        return [b - a for a, b in zip(t[:-1], t[1:])]

    def timed(nruns, nwarm):
        with torch.no_grad():
            for _ in range(nwarm):
                one()
            rows = [one() for _ in range(nruns)]
        totals = [sum(r) for r in rows]
        tot = sorted(totals)
        med = tot[len(tot) // 2]
        stages = [sorted(r[i] for r in rows)[len(rows) // 2] for i in range(3)]
        return med, stages, totals

    topo = cpu_topology()
    # n = all PHYSICAL cores of the box (torch's default pool is that already on the GPU box: 2 x 64)
    phys = topo["sockets"] * topo["cores_per_socket"]
    threads_before = torch.get_num_threads()
    if phys > 0:
        torch.set_num_threads(phys)
    threads = torch.get_num_threads()
    try:
        med, stages, totals = timed(runs, warmup)
    except BaseException:
        torch.set_num_threads(threads_before)
        raise
    # `one_frame_latency`: ONE frame on the whole box (BASELINE.md section 3's protocol, shortened to `runs` / `warmup`: the ATen CPU ops
    # of this path do not scale with threads at batch 1 — 128 threads are no faster than one).  The reported `value` is the fair host
    # number: all physical cores BUSY with independent frames (`parallel`), as the GPU path scales; it falls back to the latency
    # figure only if the parallel leg failed.
    lat = {"value": 1.0 / med, "cores": threads, "s_per_frame": med, "runs_s": [round(t, 3) for t in totals],
           "stage_s": {"backbone": stages[0], "raymarch": stages[1], "superres": stages[2]},
           "what": f"median of {runs} x synthesis(B=1) after {warmup} warm-up on {threads} intra-op threads"}
    out = {"value": lat["value"], "unit": "frames/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
           "topology": topo,
           "sample": f"{runs} x synthesis(B=1) of the {cfg.name} workload with the fp32 PyTorch-CPU oracle after {warmup} warm-up, "
                     f"{threads} threads, {med:.2f} s/frame",
           "one_frame_latency": lat}
    if check is not None and "image" in last:
        err = (check(ws, c, us, ui).float() - last["image"]).double()
        out["parity_vs_oracle"] = {"max_abs": float(err.abs().max()), "mse": float(err.pow(2).mean()),
                                   "what": "512^2 image of the HIP path (this run's default precision) vs the oracle's, same ws / "
                                           "camera / renderer uniforms, B = 1; north_star bar: MSE <= 1e-3 on [-1, 1] images"}
    try:
        if phys > 1 and os.environ.get("HFAGP_BENCH_NO_CPU_PARALLEL") != "1":
            try:
                par = out["parallel"] = cpu_parallel(cfg, state, phys)
                out["value"], out["cores"] = par["value"], par["cores"]
                out["sample"] = (f"{par['processes']} oracle processes x {par['threads_per_process']} threads, each rendering "
                                 f"{par['frames_per_process']} frames (synthesis, B=1) of the {cfg.name} workload after one warm-up "
                                 f"frame: all {par['cores']} physical cores busy, {par['wall_s']:.1f} s wall")
            except Exception as e:  # noqa: BLE001 — context only: a failed side leg must not lose the line
                out["parallel"] = {"error": f"{type(e).__name__}: {e}"}
        if n1:
            torch.set_num_threads(1)
            med1, st1, _ = timed(1, 0)
            out["n1"] = {"value": 1.0 / med1, "cores": 1, "s_per_frame": med1,
                         "stage_s": {"backbone": st1[0], "raymarch": st1[1], "superres": st1[2]}}
    finally:
        torch.set_num_threads(threads_before)        # (the caller's thread count is restored whatever happens)
    return out


class _NoGC:
    """Forward-only timed regions run without the cyclic garbage collector (see render_leg)."""

    def __enter__(self):
        gc.collect()
        gc.disable()

    def __exit__(self, *exc):
        gc.enable()


def _pct(ms_list):
    """median / p10 / p90 / max of a list of per-step milliseconds (HIP event pairs)."""
    v = sorted(ms_list)
    n = len(v)
    if n == 0:
        return None
    q = lambda f: v[min(n - 1, max(0, int(round(f * (n - 1)))))]
    return {"median": q(0.5), "p10": q(0.1), "p90": q(0.9), "max": v[-1], "n": n}


class _FitArgs:
    """Flags of code/train_3dmm.py / train_rgb.py that reach the step."""
    out_pose = False
    person_2 = False
    params_len = 76
    size = 256
    batch_size = 1
    lr = 3e-4
    latent_dim_style = 512
    latent_dim_shape = 50
    generator_seed = 0


class _AudioArgs(_FitArgs):
    """code/train_audio.py:186-212."""
    params_len = 64
    dim_aud = 64
    win_size = 16
    nosmo_iters = 0
    smo_size = 8


def train_legs(args, cfg_name, dev, rank, world, dist):
    """BASELINE configs 3 / 4 mechanics at full size: latent-basis fitting step, generator frozen, L2 loss at 256^2
    (+ the same step with the LPIPS(alex) term, random weights: `train_step_ms_lpips`), Adam 3e-4; RGB-driven (Encoder in the step: config 3) and
    3DMM-driven (config 4); ONE in-place all-reduce of the flat shared-gradient buffer per step when world > 1."""
    import torch
    from hfa_gp_amd.trainer import Trainer
    from hfa_gp_amd.synthetic import look_at_label, perturb_state
    out = {}
    # (torch.backends.cudnn.benchmark = True makes the Encoder 0.7 ms per step faster — and MIOpen's exhaustive search took
    #  3.5 minutes of a fresh box's first run: left off)

    def make(mode, lpips):
        fa = _FitArgs()
        fa.generator_preset = cfg_name
        torch.manual_seed(0)
        tr_ = Trainer(fa, dev, rank=rank, world_size=world, mode=mode, lpips=lpips)
        perturb_state(tr_.gen.generator)              # (the state of the own-size step parity tests, tests/test_gpu_round4.py)
        tr_.force_collective = os.environ.get("HFAGP_BENCH_FORCE_DIST") == "1"
        return fa, tr_

    def inputs(fa, B, g):
        real = (0.5 * torch.randn(B, 3, fa.size, fa.size, generator=g)).clamp(-1, 1).to(dev)
        params = torch.randn(B, fa.params_len, generator=g).to(dev)
        label = look_at_label(math.pi / 2 + 0.3 * torch.randn(B, generator=g),
                              math.pi / 2 + 0.155 * torch.randn(B, generator=g), flipped=False).to(dev)
        return real, params, label

    def timed(tr, fa, steps, B):
        """(max-over-ranks ms per step, {phase: mean ms} from HIP events on the launch stream)."""
        real, params, label = inputs(fa, B, torch.Generator().manual_seed(40 + rank))
        call = (lambda: tr.gen_update(real, label.clone())) if tr.mode == "rgb" else \
               (lambda: tr.gen_update(real, label.clone(), params))
        # settle: untimed steps until two consecutive ones agree to 2 % (at least 2, at most 10) — a new trainer's first steps grow the
        # caching allocator, build weight images and (RGB) run the Encoder's first-use paths; round 5's first line timed the RGB step
        # at 16.9 ms with a 8.3 ms forward phase after two warm-up steps, 12.6 - 12.8 ms in every longer run (fit_rgb, dev scripts)
        prev = None
        for i in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call()
            e1.record()
            torch.cuda.synchronize()
            cur = e0.elapsed_time(e1)
            if i >= 1 and prev is not None and abs(cur - prev) <= 0.02 * prev:
                break
            prev = cur
        # the few timed steps run with the cyclic collector off, like the render leg (a generation-2 collection is ~50 ms of host
        # time: in a 10-step region it is 3 - 5 ms per step of noise); the 250-step fitting legs below keep it ON
        gc.collect()
        gc_was = gc.isenabled()
        gc.disable()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        tr.timing = {}
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        evs[0].record()
        t0 = time.perf_counter()
        for i in range(steps):
            res = call()
            evs[i + 1].record()
        torch.cuda.synchronize()
        if gc_was:
            gc.enable()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        spans, tr.timing = tr.timing, None
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        assert torch.isfinite(res[-3]), "fitting step produced a non-finite loss"
        phases = {k: sum(a.elapsed_time(b) for a, b in v) / max(len(v), 1) for k, v in spans.items()}
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
        phases["step_events_median"] = per[len(per) // 2]        # per-step HIP event pairs on the launch stream (this rank)
        phases["step_events_max"] = per[-1]
        return dt / steps * 1e3, phases

    def kernel_events(tr, fa, B, steps=3):
        """A SECOND pass of the same step with HIP events around the kernel families that carry it (generator.timing; the
        events cost host time per launch, so the timed pass runs without them): {key: (ms per step, units per step, launches
        per step)} — what `roofline_train` is computed from."""
        real, params, label = inputs(fa, B, torch.Generator().manual_seed(40 + rank))
        g = tr.gen.generator
        g.timing = {}
        for _ in range(steps):
            if tr.mode == "rgb":
                tr.gen_update(real, label.clone())
            else:
                tr.gen_update(real, label.clone(), params)
        torch.cuda.synchronize()
        table, g.timing = g.timing, None
        return {k: (sum(e0.elapsed_time(e1) for e0, e1, _ in v) / steps, sum(u for _, _, u in v) / steps, len(v) / steps)
                for k, v in table.items()}

    B = args.train_batch
    batches = [B] if (args.no_sweep or world > 1) else sorted({1, B, 4})
    for mode in ("3dmm", "rgb"):
        fa, tr = make(mode, "none")
        leg = {"frozen": {}, "tuned": {}, "shared_grad_bytes": None}
        for b in (batches if mode == "3dmm" else [B]):
            ms, phases = timed(tr, fa, args.train_steps, b)
            leg["frozen"][b] = {"step_ms": ms, "phases_ms": phases}
        leg["shared_grad_bytes"] = 4 * tr.flat_grads().numel
        leg["kernel_events"] = kernel_events(tr, fa, B)
        # after tune_iter the reference also trains the generator (trainer_rgb.py:69-71): all 30.7 M parameters get
        # gradients, which live in the same flat buffer and are all-reduced with the basis / driver gradients
        tr.tune_generator()
        ms, phases = timed(tr, fa, max(2, args.train_steps // 2), B)
        leg["tuned"][B] = {"step_ms": ms, "phases_ms": phases}
        leg["shared_grad_bytes_tuned"] = 4 * tr.flat_grads().numel
        leg["kernel_events_tuned"] = kernel_events(tr, fa, B)
        out[mode] = leg
        del tr
        torch.cuda.empty_cache()
    if args.lpips and world == 1:
        from hfa_gp_amd.lpips_alex import LPIPSAlex
        torch.manual_seed(1234)               # seeded random weights
        fa, tr = make("rgb", LPIPSAlex().to(dev))
        ms, phases = timed(tr, fa, args.train_steps, B)
        out["rgb_lpips"] = {"step_ms": ms, "phases_ms": phases,
                            "note": "LPIPS(alex) architecture with RANDOM weights: the cost of the reference objective "
                                    "l2 + lpips (trainer_rgb.py:86-91), not its value"}
        del tr
        torch.cuda.empty_cache()
    return out, B


def fit3dmm_leg(args, cfg_name, dev, rank, world, dist):
    """BASELINE config 4 as written: 3DMM-driven fitting (`train_3dmm.py:85-128`) of 250 x world synthetic frames (8 GPUs:
    the 2000 frames of the config) sharded in contiguous blocks — rank r owns [250 r, 250 (r + 1)) — batch 2 per rank,
    one in-place all-reduce of the flat shared-gradient buffer per step; one pass."""
    import torch
    from hfa_gp_amd.synthetic import make_frame_set
    from hfa_gp_amd.trainer import Trainer, fit_frames, shard_range
    fa = _FitArgs()
    fa.generator_preset = cfg_name
    fa.batch_size = args.train_batch * world
    n = args.fit3dmm_frames_per_rank * world
    torch.manual_seed(3)
    tr = Trainer(fa, dev, rank=rank, world_size=world, mode="3dmm", lpips="none")
    tr.force_collective = os.environ.get("HFAGP_BENCH_FORCE_DIST") == "1"
    # every rank renders only its own shard of the targets (the data set is synthetic: same seed -> same frames)
    lo, hi = shard_range(n, rank, world)
    full = make_frame_set(tr.gen, n, size=fa.size, seed=44, params_len=fa.params_len, only=(lo, hi))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    gc.collect()                       # (before the clock starts; the pass itself keeps the collector ON: a training loop's cyclic
    #                                    garbage holds device memory, and with the collector off the 250 steps ran 0.4-0.7 ms slower)
    t0 = time.perf_counter()
    marks = {}

    def on_step(i, _out):
        if i in (half, last):
            marks[i] = torch.cuda.Event(enable_timing=True)
            marks[i].record()
    from hfa_gp_amd.trainer import epoch_batches
    nsteps = sum(1 for _ in epoch_batches(n, rank, world, args.train_batch))
    half, last = nsteps // 2, nsteps - 1
    losses = fit_frames(tr, full["real"], full["label"], full["params"], epochs=1, batch=args.train_batch, on_step=on_step)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    l = torch.stack(losses).float().cpu()
    k = max(1, len(l) // 10)
    return {"workload": f"train_3dmm-style fit, {n} synthetic frames = {args.fit3dmm_frames_per_rank} per rank in contiguous "
                        f"blocks (rank r owns [{args.fit3dmm_frames_per_rank} r, {args.fit3dmm_frames_per_rank} (r+1))), "
                        f"Weights_3DMM -> basis -> generator, L2, Adam 3e-4, batch {args.train_batch}/rank, one pass",
            "steps": len(l), "ms_per_step": dt / max(len(l), 1) * 1e3,
            "ms_per_step_second_half": (marks[half].elapsed_time(marks[last]) / max(last - half, 1)
                                        if half in marks and last in marks and last > half else None),
            "frames_per_s": n / dt,
            "loss_first_tenth": float(l[:k].mean()), "loss_last_tenth": float(l[-k:].mean())}


def fit_leg(args, cfg_name, dev, rank, world, dist):
    """BASELINE config 3 as written: RGB-driven latent-basis fitting (`train_rgb.py:114-154`) on `--fit-frames`
    synthetic frames rendered from a hidden true basis (hfa_gp_amd.synthetic), frames sharded in contiguous blocks over
    the ranks, one pass, batch 2 per rank, L2 loss; reports ms/step and the loss at both ends of the pass."""
    import torch
    from hfa_gp_amd.synthetic import make_frame_set
    from hfa_gp_amd.trainer import Trainer, fit_frames
    fa = _FitArgs()
    fa.generator_preset = cfg_name
    fa.batch_size = args.train_batch * world
    torch.manual_seed(1)
    tr = Trainer(fa, dev, rank=rank, world_size=world, mode="rgb", lpips="none")
    data = make_frame_set(tr.gen, args.fit_frames, size=fa.size, seed=40)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    gc.collect()                       # (before the clock starts; the pass itself keeps the collector ON: a training loop's cyclic
    #                                    garbage holds device memory, and with the collector off the 250 steps ran 0.4-0.7 ms slower)
    t0 = time.perf_counter()
    marks = {}          # HIP events on the launch stream at the middle and the end of the pass: the steady-state step time

    def on_step(i, _out):
        if i in (half, last):
            marks[i] = torch.cuda.Event(enable_timing=True)
            marks[i].record()
    from hfa_gp_amd.trainer import epoch_batches
    nsteps = sum(1 for _ in epoch_batches(args.fit_frames, rank, world, args.train_batch))
    half, last = nsteps // 2, nsteps - 1
    losses = fit_frames(tr, data["real"], data["label"], epochs=1, batch=args.train_batch, on_step=on_step)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    l = torch.stack(losses).float().cpu()
    k = max(1, len(l) // 10)
    steady = marks[half].elapsed_time(marks[last]) / max(last - half, 1) if half in marks and last in marks and last > half else None
    return {"workload": f"train_rgb-style fit, {args.fit_frames} synthetic frames (targets rendered from a hidden basis, "
                        f"256^2), Encoder -> basis -> generator, L2, Adam 3e-4, batch {args.train_batch}/rank, one pass",
            "steps": len(l), "ms_per_step": dt / max(len(l), 1) * 1e3, "ms_per_step_second_half": steady,
            "frames_per_s": args.fit_frames / dt,
            "loss_first_tenth": float(l[:k].mean()), "loss_last_tenth": float(l[-k:].mean())}


def audio_leg(args, cfg_name, dev, rank, world, dist):
    """BASELINE config 5: batched audio-driven reenactment (`run_recon_video_audio.py:346`, `trainer_audio.py:115-153`)
    at 512^2 / 96 samples with the super-resolution blocks on the single-pass fp16 MFMA path (the reference's own CUDA
    precision split); `--audio-frames` frames sharded in contiguous blocks over the ranks, B = --batch per synthesis."""
    import torch
    from hfa_gp_amd.synthetic import audio_features, gaussian_labels
    from hfa_gp_amd.trainer import AudioTrainer, shard_range
    fa = _AudioArgs()
    fa.generator_preset = cfg_name
    n = args.audio_frames
    torch.manual_seed(2)
    tr = AudioTrainer(audio_features(n).numpy(), n, fa, dev, rank=rank, world_size=world, lpips="none")
    tr.gen.generator.sr_conv_precision = "f16"
    tr.gen.generator.sr_storage = "f16"          # forward-only reenactment: fp16 tensors between the SR layers, as EG3D's
    lo, hi = shard_range(n, rank, world)
    labels = gaussian_labels(n, dev, seed=51)
    idx = torch.arange(lo, hi, device=dev)
    B = args.batch

    def run():
        for i in range(0, hi - lo, B):
            tr.sample_frames(idx[i:i + B], labels[lo + i: lo + i + B].clone())
    run()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    with _NoGC():                              # (entered BEFORE the clock starts: the collection itself takes ~50 ms)
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return {"workload": f"audio-driven batched reenactment, {n} frames of aud[N,16,29], smoothing window 8, AudioNet + "
                        f"AudioAttNet -> basis -> generator, 512^2 @ 48+48 samples, SR convs fp16 MFMA, B={B}",
            "frames_per_s": n / dt, "frames_per_s_per_gpu": n / dt / world, "ms_per_frame": dt / max(hi - lo, 1) * 1e3}


def main():
    args = parse()
    # The LPIPS leg is the only MIOpen user of this run (AlexNet convs on PyTorch-ROCm).  On a fresh box MIOpen's default
    # find mode benchmarks every new conv shape on first use: 45 s of the 49 s that leg took.  FAST mode picks the kernel
    # from the find-db / heuristics without timing candidates; nothing in the hot path goes through MIOpen.
    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
    # ... and its per-solver "workspace required" warnings (hundreds of lines) would be all the driver's output tail holds
    os.environ.setdefault("MIOPEN_LOG_LEVEL", "3")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the N ranks ourselves, exactly as the driver does
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("NCCL_DEBUG", "WARN")          # RCCL's own warnings (ring build-up, IPC) reach stderr of the run
        port = _free_port()
        sys.stderr.write(f"[bench] no launcher (WORLD_SIZE unset): starting {args.gpus} ranks under torch.distributed.run, "
                         f"rendezvous 127.0.0.1:{port}\n")
        sys.stderr.flush()
        sys.exit(subprocess.run(launcher_command(args.gpus, sys.argv[1:], port), env=env).returncode)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py needs an MI355X (torch.cuda.is_available() is False); no CPU fallback [rank {rank} of {world}]")
    if args.share_device:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} but only {torch.cuda.device_count()} device(s) visible "
                         f"(developer check of the N > 1 path on one GPU: --backend gloo --share-device)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # The contract is ONE JSON line on stdout.  Native libraries write banners straight to file descriptor 1 (gloo: "[Gloo]
    # Rank 0 is connected to ..."; RCCL / MIOpen warnings), so everything between here and the final print goes to stderr:
    # fd 1 is pointed at fd 2 and restored for the one line rank 0 prints.
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    dist = None
    # (HFAGP_BENCH_FORCE_DIST=1, developer: a ONE-rank process group, so that a 1-GPU box runs every collective call of the
    # N > 1 code path — RCCL init with device_id, barriers, the MAX reduction of the timing, the trainers' all-reduces)
    if world > 1 or os.environ.get("HFAGP_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        # RCCL's log of THIS run goes to a per-rank file (fd 1 / 2 stay readable): INFO level with the INIT and TUNING subsystems, so
        # that the line can say which algorithm / protocol RCCL picked for each all-reduce payload (`allreduce_us.algo`: the
        # 1.46 MB / 6.98 MB / 92 MB / 123 MB buffers are latency- resp. bandwidth-bound on 7 x 153 GB/s xGMI links — ring is 14 hops,
        # SURVEY section 5.8).  A caller's own NCCL_DEBUG* settings win.
        if args.backend == "nccl":
            os.environ.setdefault("NCCL_DEBUG", "INFO")
            os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,TUNING")
            os.environ.setdefault("NCCL_DEBUG_FILE", f"/tmp/hfagp_rccl_rank{rank}_{os.getpid()}.log")
        import datetime
        # a rank that dies (or never starts) must end the job with a message, not hang its peers for the default 10 - 30 min
        tmo = datetime.timedelta(seconds=args.dist_timeout)
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
            if dist.get_world_size() != args.gpus:
                raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")
            # RCCL builds its rings at the FIRST collective: do it here, where a failure can be named, and check the answer
            probe = torch.full((1,), float(rank + 1), device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            if abs(float(probe) - world * (world + 1) / 2) > 1e-3:
                raise RuntimeError(f"first all-reduce returned {float(probe)} instead of {world * (world + 1) / 2}")
        except SystemExit:
            raise
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"[bench rank {rank}/{world} on cuda:{local_rank}] could not bring up the {args.backend} process "
                             f"group within {args.dist_timeout} s: {type(e).__name__}: {e}\n"
                             f"  (one process per GPU, rendezvous 127.0.0.1:{os.environ.get('MASTER_PORT')}; RCCL needs "
                             f"HSA_ENABLE_IPC_MODE_LEGACY=0 on this host driver; every rank must see all {world} devices — do not "
                             f"narrow HIP_VISIBLE_DEVICES per rank, the rank picks cuda:LOCAL_RANK itself)\n")
            raise SystemExit(3)
    n_ranks = dist.get_world_size() if dist is not None else 1

    from hfa_gp_amd.config import PRESETS
    from hfa_gp_amd.generator import TriPlaneGenerator
    from hfa_gp_amd.synthetic import make_inputs, perturb_state, state_cpu

    cfg = PRESETS[args.preset]()
    # the generator the parity tests run the oracle against at this batch (tests/test_gpu_round3.py::benched): EG3D init
    # + non-zero biases and noise strengths, so that the timed state is the tested state
    gen = perturb_state(TriPlaneGenerator(cfg, seed=0)).requires_grad_(False)
    state = state_cpu(gen) if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    gen = gen.to(dev)
    B = args.batch
    # each rank renders its own frames (frame-parallel): different seed per rank
    ws, c, us, ui = make_inputs(cfg, B, seed=10 + rank)
    ws, c, us, ui = ws.to(dev), c.to(dev), us.to(dev), ui.to(dev)

    def step():
        # the renderer's stratified / importance uniforms are DRAWN INSIDE the timed call, as EG3D's ImportanceRenderer does on every
        # synthesis (torch.rand on the device: 50 M floats at B = 32); the parity legs hand over fixed ones
        return gen.synthesis(ws, c, noise_mode="const")["image"]

    def render_leg(precision, sr_precision=None, events=True, sr_storage="f32"):
        """W warm-up + K timed steps with the conv GEMMs in ``precision`` (super-resolution blocks: ``sr_precision``
        when given); (seconds max over ranks, this rank's seconds, event table)."""
        gen.conv_precision = precision
        gen.sr_conv_precision = sr_precision
        gen.sr_storage = sr_storage
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        gen.timing = {} if events else None
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        # (as `timeit` does: no cyclic-GC pass inside the timed region — a generation-2 collection over the heap this process
        # has built by now takes ~50 ms, and one landed in the 20 timed steps of two of three full runs: 754 frames/s by the
        # wall clock against 806 by the per-step HIP events of the same steps)
        gc.collect()
        gc.disable()
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(args.steps):
            img = step()
            marks[i + 1].record()
        torch.cuda.synchronize()
        own = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gc.enable()
        timing, gen.timing = (gen.timing or {}), None
        timing["step"] = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        assert torch.isfinite(img).all()
        return dt, own, timing

    def sweep_leg(b, precision, steps=20, warmup=3):
        """batch sizes 1, 4 (and 16) beside the headline's 8; per-step HIP event pairs."""
        gen.conv_precision, gen.sr_conv_precision = precision, None
        if b > 32:
            torch.cuda.empty_cache()        # (the large batches want their activations contiguous: drop the other legs' cache)
        w_, c_, us_, ui_ = [t.to(dev) for t in make_inputs(cfg, b, seed=10 + rank)]
        for _ in range(warmup):
            gen.synthesis(w_, c_, noise_mode="const", u_strat=us_, u_imp=ui_)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        ev[0].record()
        for i in range(steps):
            gen.synthesis(w_, c_, noise_mode="const", u_strat=us_, u_imp=ui_)
            ev[i + 1].record()
        torch.cuda.synchronize()
        p = _pct([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)])
        p["frames_per_s_per_gpu"] = b / (p["median"] * 1e-3)
        return p

    leg_s = {}                 # wall seconds per leg of this script (rank 0), for whoever budgets the run

    def timed_leg(name, fn, *a):
        t = time.perf_counter()
        try:
            out_ = fn(*a)
            torch.cuda.synchronize()
        except BaseException as e:  # noqa: BLE001 — name the rank and device, then fail the run (a collective that lost a peer
            # surfaces here as a timeout on EVERY surviving rank: the first line in the log names the one that broke)
            sys.stderr.write(f"[bench rank {rank}/{world} on cuda:{local_rank}, pid {os.getpid()}] leg '{name}' failed: "
                             f"{type(e).__name__}: {e}\n")
            sys.stderr.flush()
            raise
        leg_s[name] = round(leg_s.get(name, 0.0) + time.perf_counter() - t, 2)
        return out_

    prec = args.precision or cfg.conv_precision
    # Before anything is timed: bring the device out of its idle power state (the process has just spent ~10 s importing
    # torch) and let the caching allocator reach its steady state — up to 12 untimed steps, stopping once two consecutive
    # step times agree to 1 %.  One run taken right after the GPU test suite lost 58 ms of its 20 timed steps to such a
    # one-off stall (wall 42.1 ms/step against 39.2 ms/step by HIP events).  The contract's W warm-up steps follow as asked.
    def settle():
        gen.conv_precision, gen.sr_conv_precision, gen.sr_storage = prec, None, "f32"
        prev, n_settled = None, 0
        for _ in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            step()
            e1.record()
            torch.cuda.synchronize()
            cur = e0.elapsed_time(e1)
            # (at least six steps: two agreeing steps right after start-up have been followed by a one-off 60 ms stall inside the timed
            # region — a run taken straight after the GPU test suite, round 6: 95.8 ms for one step of 35.5)
            if n_settled >= 5 and prev is not None and abs(cur - prev) <= 0.01 * prev:
                break
            prev = cur
            n_settled += 1
    timed_leg("settle", settle)
    # the headline leg first, exactly as the contract words it (W warm-up steps, then K timed steps), WITHOUT the
    # per-kernel timing events (they cost host time per launch); a second pass of the same leg collects the events
    # the roofline objects are computed from
    dt, own_dt, timing_head = timed_leg("render", render_leg, prec, None, False)
    _, _, timing = timed_leg("render", render_leg, prec)
    timing["step"] = timing_head["step"]
    dt32 = timing32 = dtb3 = None
    # single-GPU diagnostics (other precisions, batch sweeps, the LPIPS-cost leg, the CPU baseline) are skipped at N > 1: a
    # multi-GPU run measures the headline, the fitting legs with their collectives and the sharded fits — nothing else
    diag = n_ranks == 1
    if prec != "fp32" and not args.no_fp32_leg and diag:
        dt32, _, timing32 = timed_leg("render_other_precisions", render_leg, "fp32")
        if prec != "bf16x3":
            dtb3, _, _ = timed_leg("render_other_precisions", render_leg, "bf16x3", None, False)
    dtx2 = timingx2 = None
    if prec != "f16x2" and not args.no_f16_leg and diag:
        # TF32 class (not the headline): 22-bit weights x activations rounded to one fp16 part, 2 MFMAs per product
        dtx2, _, timingx2 = timed_leg("render_other_precisions", render_leg, "f16x2")
    dt16sr = dt16 = dt16srh = timing16 = None
    if not args.no_f16_leg and diag:
        # the reference's CUDA defaults: fp32-class backbone, fp16 super-resolution; then every conv in fp16
        dt16sr, _, timing16 = timed_leg("render_other_precisions", render_leg, prec, "f16")
        dt16, _, _ = timed_leg("render_other_precisions", render_leg, "f16", None, False)
        # ... and with the super-resolution activations STORED in fp16 as well (EG3D's fp16 blocks; forward-only calls)
        dt16srh, _, _ = timed_leg("render_other_precisions", render_leg, prec, "f16", False, "f16")
        gen.sr_storage = "f32"
    sweep = None
    if not args.no_sweep and diag:
        # (64 and 128 frames per call: what the 288 GB of HBM allow beyond the headline's batch — not the headline, whose batch
        # is the one the parity tests run the oracle at)
        sweep = {}
        for b in (1, 4, 8, 16, 32, 64, 128):
            if b == B:
                continue
            try:
                sweep[str(b)] = timed_leg("batch_sweep", sweep_leg, b, prec, 20 if b <= 16 else 10 if b <= 32 else 4, 3 if b <= 32 else 2)
            except torch.cuda.OutOfMemoryError:          # (a batch beyond the headline's that does not fit this device: leave it out)
                torch.cuda.empty_cache()
                sweep[str(b)] = {"skipped": "out of memory"}
    gen.conv_precision, gen.sr_conv_precision = prec, None
    overflow = gen.f16_range_report()

    def agg(key, table=None):
        evs = (timing if table is None else table).get(key, [])
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in evs)
        units = sum(u for _, _, u in evs)
        return ms, units, len(evs)

    per_rank = [B * args.steps / own_dt]
    if dist is not None:
        t = torch.zeros(n_ranks, device=dev, dtype=torch.float64)
        t[rank] = B * args.steps / own_dt
        dist.all_reduce(t)
        per_rank = [float(v) for v in t.cpu()]

    del ws, c, us, ui
    train = train_B = fit = fit3 = audio = None
    if not args.no_train:
        torch.cuda.empty_cache()
        train, train_B = timed_leg("train_steps", train_legs, args, args.preset, dev, rank, world, dist)
        if args.fit_frames > 0:
            fit = timed_leg("fit_rgb", fit_leg, args, args.preset, dev, rank, world, dist)
            torch.cuda.empty_cache()
        if args.fit3dmm_frames_per_rank > 0:
            fit3 = timed_leg("fit_3dmm_sharded", fit3dmm_leg, args, args.preset, dev, rank, world, dist)
    if args.audio_frames > 0:
        torch.cuda.empty_cache()
        audio = timed_leg("audio_reenactment", audio_leg, args, args.preset, dev, rank, world, dist)

    def profiled_traffic(prefix):
        """HBM bytes per launch from the committed PMC passes (profiles/traffic.json), only if the batch matches."""
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if t.get("batch") != B:
                return None
            for k, v in t.items():
                if k.startswith(prefix):
                    return v["hbm_bytes"]
        except Exception:
            pass
        return None

    if rank == 0:
        frames = n_ranks * B * args.steps
        rm_ms, rm_bytes, rm_n = agg("raymarch")
        rm_gbs = rm_bytes / (rm_ms * 1e-3) / 1e9

        def f32_roofline(table):
            ms, flops, n = agg("modconv", table)
            ms_u, flops_u, n_u = agg("modconv_up", table)
            ms, flops, n = ms + ms_u, flops + flops_u, n + n_u
            tf = flops / (ms * 1e-3) / 1e12
            return {"bound": "mfma", "kernel": "modconv_kernel (v_mfma_f32_32x32x2_f32, exact fp32)",
                    "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
                    "traffic": profiled_traffic("modconv_kernel<2, 2, 2, 2>"), "avg_launch_ms": ms / max(n, 1),
                    "launches": n}

        if prec == "fp32":
            roof = f32_roofline(timing)
        else:
            # split-operand path: ALGORITHMIC flops (2*M*N*K of the fp32 conv) against the 16-bit dense MFMA peak (2.5 PF
            # for f16 and bf16 alike) divided by the MFMAs each algorithmic product costs (3 or 6)
            # The DOMINANT kernel is the 9-tap instance (the nine 3x3 layers of a synthesis: half of the step); the merged
            # up-sampling conv (eight layers) and both together are reported beside it.
            ms, flops, n = agg("modconv_split")
            ms_up, flops_up, n_up = agg("modconv_split_up")
            tf = flops / (ms * 1e-3) / 1e12
            peak = MFMA_BF16_PEAK_TFLOPS / SPLIT_MFMAS[prec]
            kd = {"f16x3": 4, "bf16x3": 2, "bf16x6": 3}[prec]
            roof = {"bound": "mfma", "kernel": f"modconv_bf16_kernel<{kd}, 2, 9> ({SPLIT_MFMAS[prec]} x "
                                               f"v_mfma_f32_32x32x16_{SPLIT_ELEM[prec]} per fp32 product)",
                    "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                    "traffic": profiled_traffic(f"modconv_bf16_kernel<{kd}, 2, 9"), "avg_launch_ms": ms / max(n, 1),
                    "launches": n, "mfma_16bit_tflops": tf * SPLIT_MFMAS[prec],
                    "traffic_source": "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of THIS command taken on the "
                                      "GPU box when the kernels last changed (profiles/run_profile.sh), not measured during this run"}
            if n_up:
                tf_up = flops_up / (ms_up * 1e-3) / 1e12
                tf_all = (flops + flops_up) / ((ms + ms_up) * 1e-3) / 1e12
                roof["up_conv"] = {"kernel": f"upconv_bf16_kernel<{kd}> (stride-2 transposed 3x3, four phases merged; flops counted "
                                             "at INPUT resolution)", "achieved": tf_up, "frac": tf_up / peak,
                                   "avg_launch_ms": ms_up / n_up, "launches": n_up}
                roof["all_conv_gemms"] = {"achieved": tf_all, "frac": tf_all / peak}
            # every up-sampling LAYER as a whole: transposed-conv GEMM + FIR epilogue (or the fused kernel + its strip
            # fix-ups), algorithmic flops at input resolution against the same ceiling
            ups = []
            for kname in sorted(timing):
                if kname.startswith("up_layer:"):
                    ms_l, fl_l, n_l = agg(kname)
                    if n_l:
                        ups.append({"layer": kname[len("up_layer:"):], "ms_per_launch": ms_l / n_l, "achieved": fl_l / (ms_l * 1e-3) / 1e12,
                                    "frac": fl_l / (ms_l * 1e-3) / 1e12 / peak, "launches": n_l})
            if ups:
                roof["up_layers"] = ups
            # the up-sampling LAYER fused with its FIR + epilogue (csrc/upconv_fir.hip; taken for short-K layers: the first
            # super-resolution layer): the whole layer is one launch (+ two strip fix-ups), so this is the layer's rate
            ms_uf, flops_uf, n_uf = agg("modconv_split_upfir")
            if n_uf:
                roof["up_layer_fused"] = {"kernel": f"upfir_lean_kernel<{kd}> (Cin = 32: transposed conv on 16x16x32 MFMAs with swapped "
                                                    "operands, FIR in registers + DPP, demod / noise / bias / act, one pass, no scratch)",
                                          "achieved": flops_uf / (ms_uf * 1e-3) / 1e12,
                                          "frac": flops_uf / (ms_uf * 1e-3) / 1e12 / peak, "avg_launch_ms": ms_uf / n_uf,
                                          "launches": n_uf}
        # ray march: the planes of a frame (25 MB) are cache resident, so the SURVEY 8d "algorithmic bytes" are a GATHER
        # rate served by L2 / Infinity Cache, not HBM traffic.  The kernel's physical floor is the L2 gather
        # (gather bytes / 34.5 TB/s) plus the decoder MLP on the fp32 matrix pipe (13.1 GF per frame / 157.3 TF); `frac`
        # is that floor over the measured time.  HBM traffic from the counters and its fraction of 8 TB/s are listed
        # separately and are NOT the roofline fraction.
        s_tot = cfg.depth_resolution + cfg.depth_resolution_importance
        r = cfg.neural_rendering_resolution ** 2
        frames_per_launch = B
        gather_bytes = frames_per_launch * r * s_tot * 3 * 4 * 32 * 4
        dec_flops = frames_per_launch * r * s_tot * 2.0 * (32 * 64 + 64 * 33)
        rm_avg_ms = rm_ms / max(rm_n, 1)
        # decoder MLP: split fp16 operands on the 16-bit pipe (3 MFMAs per product against 2.5 PF dense) or exact fp32 MFMA
        dec16 = cfg.decoder_precision == "f16x3"
        dec_peak = MFMA_F16_PEAK_TFLOPS / 3 if dec16 else MFMA_F32_PEAK_TFLOPS
        floor_ms = (gather_bytes / (L2_PEAK_GBS * 1e9) + dec_flops / (dec_peak * 1e12)) * 1e3
        rm_traffic = profiled_traffic("raymarch_kernel")
        compulsory = frames_per_launch * (3 * cfg.plane_resolution ** 2 * 32 * 4 + r * (34 + s_tot) * 4)
        out = {
            "metric": "rendered 512^2 frames/sec (96 depth samples), whole job",
            "value": frames / dt, "unit": "frames/s", "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if prec == "fp32" else f"f32 tensors and accumulation; conv GEMM products as {prec} "
                                                  f"(operands split into {SPLIT_ELEM[prec]} parts, {SPLIT_MFMAS[prec]} "
                                                  f"MFMAs per product" + (": 22 mantissa bits, fp32-class results)"
                                                                          if prec == "f16x3" else ")"),
            "data": "synthetic",
            "renderer_uniforms": "drawn inside the timed call (torch.rand on the device per synthesis, as EG3D's ImportanceRenderer does)",
            "config": {"workload": f"{cfg.name}: synthesis(ws[B,14,512], c[B,25]) forward, 512x512 out, 128^2 rays x "
                                   f"(48+48) samples, random-init weights, random latents+cameras",
                       "frames_per_step_per_gpu": B, "parallelism": f"frame-parallel x{n_ranks}"},
            "per_rank_frames_per_s": per_rank,
            # (max - min) / median over the ranks: frame-parallel rendering has no collective in the timed region, so a spread
            # beyond a few per cent means one GPU (clock, thermal state, a noisy neighbour on its NUMA node) holds the job back
            "per_rank_spread": ((max(per_rank) - min(per_rank)) / sorted(per_rank)[len(per_rank) // 2]) if per_rank else 0.0,
            # dominant kernel by time: the modulated-conv implicit GEMM
            "roofline": roof,
            # the kernel north_star names: see the comment above
            "roofline_raymarch": {"bound": "l2+mfma", "kernel": "raymarch_kernel<3,3> (decoder: " + cfg.decoder_precision + ")",
                                  "achieved": frames_per_launch / (rm_avg_ms * 1e-3),
                                  "peak": frames_per_launch / (floor_ms * 1e-3), "unit": "frames/s (kernel alone)",
                                  "frac": floor_ms / rm_avg_ms,
                                  "floor_ms_per_launch": {"l2_gather": gather_bytes / (L2_PEAK_GBS * 1e9) * 1e3,
                                                          ("decoder_mfma_f16x3" if dec16 else "decoder_mfma_f32"):
                                                              dec_flops / (dec_peak * 1e12) * 1e3},
                                  "avg_launch_ms": rm_avg_ms, "launches": rm_n,
                                  "gather_rate_GBps_survey8d": rm_gbs,
                                  "gather_rate_over_hbm_peak_survey8d": rm_gbs / HBM_PEAK_GBS,
                                  "algorithmic_bytes_per_launch": rm_bytes / max(rm_n, 1),
                                  "compulsory_bytes_per_launch": compulsory,
                                  "traffic": rm_traffic,
                                  "hbm_counter_frac_of_peak": (rm_traffic / (rm_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                                                               if rm_traffic else None),
                                  "traffic_over_compulsory": rm_traffic / compulsory if rm_traffic else None},
        }
        out["config"]["conv_precision"] = prec
        if overflow is not None:
            # fp16 range tracking: largest |activation| any unclamped layer produced in the last frame batch; the fp16-part
            # GEMMs normalise by it (exact power of two), no layer fell back to another precision
            out["f16_range"] = {"max_abs_activation": max(overflow.values()), "fp16_max": 65504.0,
                                "layers_fell_back": []}
        if dt16sr is not None:
            # not the headline: products of the super-resolution convs (value_f16_sr: the reference's own CUDA
            # precision split) or of all convs (value_f16) rounded to fp16, one fp16 MFMA per product
            out["value_f16_sr"] = frames / dt16sr
            out["value_f16"] = frames / dt16
            out["value_f16_sr_f16_storage"] = frames / dt16srh     # + activations between the SR layers kept in fp16
            ms, flops, n = agg("modconv_f16", timing16)
            ms_u, flops_u, n_u = agg("modconv_f16_up", timing16)
            ms, flops, n = ms + ms_u, flops + flops_u, n + n_u
            tf = flops / (ms * 1e-3) / 1e12
            out["roofline_f16_sr"] = {"bound": "mfma", "kernel": "modconv_bf16_kernel<1> / upconv_bf16_kernel<1> "
                                                                 "(1 x v_mfma_f32_32x32x16_f16 per product)",
                                      "achieved": tf, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": tf / MFMA_F16_PEAK_TFLOPS, "traffic": None,
                                      "avg_launch_ms": ms / max(n, 1), "launches": n}
        if dtb3 is not None:
            out["value_bf16x3"] = frames / dtb3      # bf16 hi+lo parts: ~5 % faster, product error 2^-16 instead of 2^-22
        if dtx2 is not None:
            # not the headline: the arithmetic CLASS of the reference's own GPU path (its scripts leave cuDNN's TF32 on:
            # train_rgb.py:13-14) — 11-bit activations x 22-bit weights, fp32 accumulation; image MSE vs the oracle ~1e-7
            ms, flops, n = agg("modconv_split", timingx2)
            tf = flops / (ms * 1e-3) / 1e12
            out["tf32_class_leg"] = {"conv_precision": "f16x2", "value": frames / dtx2, "unit": "frames/s",
                                     "mfma_per_product": 2,
                                     "conv3x3": {"achieved": tf, "peak": MFMA_F16_PEAK_TFLOPS / 2, "unit": "TFLOP/s",
                                                 "frac": tf / (MFMA_F16_PEAK_TFLOPS / 2), "avg_launch_ms": ms / max(n, 1),
                                                 "launches": n},
                                     "note": "opt-in (conv_precision='f16x2'); the headline stays f16x3 (fp32 class)"}
        if dt32 is not None:
            out["value_fp32_exact"] = frames / dt32
            out["roofline_fp32_exact"] = f32_roofline(timing32)
        out["step_ms"] = _pct(timing["step"])       # per-step HIP event pairs of the timed region (this rank)
        out["step_ms"]["argmax"] = max(range(len(timing["step"])), key=lambda i: timing["step"][i])      # (which step a stall hit)
        if sweep is not None:
            out["batch_sweep"] = sweep
        if train is not None:
            t3, trgb = train["3dmm"], train["rgb"]
            out["train_step_ms"] = trgb["frozen"][train_B]["step_ms"]             # BASELINE config 3: RGB-driven
            out["train_step_ms_3dmm"] = t3["frozen"][train_B]["step_ms"]          # BASELINE config 4 mechanics
            out["train_step_ms_generator_tuned"] = trgb["tuned"][train_B]["step_ms"]
            out["train_step_ms_3dmm_generator_tuned"] = t3["tuned"][train_B]["step_ms"]
            # fwd / bwd / all-reduce / Adam from HIP events on the launch stream (mean ms per step, this rank)
            out["train_phases_ms"] = trgb["frozen"][train_B]["phases_ms"]
            out["train_phases_ms_3dmm"] = t3["frozen"][train_B]["phases_ms"]
            out["train_phases_ms_generator_tuned"] = trgb["tuned"][train_B]["phases_ms"]
            out["train_step_ms_3dmm_by_batch"] = {str(b): v["step_ms"] for b, v in t3["frozen"].items()}
            out["allreduce_us"] = {"rgb": 1e3 * trgb["frozen"][train_B]["phases_ms"].get("allreduce", 0.0),
                                   "3dmm": 1e3 * t3["frozen"][train_B]["phases_ms"].get("allreduce", 0.0),
                                   "rgb_generator_tuned": 1e3 * trgb["tuned"][train_B]["phases_ms"].get("allreduce", 0.0),
                                   "bytes": {"rgb": trgb["shared_grad_bytes"], "3dmm": t3["shared_grad_bytes"],
                                             "rgb_generator_tuned": trgb["shared_grad_bytes_tuned"]}}
            out["allreduce_us"]["algo"] = rccl_choices(os.environ.get("NCCL_DEBUG_FILE"), n_ranks) if dist is not None else None
            out["train_config"] = {"workload": "latent-basis fitting step (fwd + bwd + Adam), K=50, generator frozen, L2 at "
                                               "256^2, synthetic frames; train_step_ms = RGB-driven (Encoder(256) in the "
                                               "step, trainer_rgb.py:73-98), train_step_ms_3dmm = 3DMM-driven "
                                               "(trainer_3dmm.py:43-67)",
                                   "frames_per_step_per_gpu": train_B,
                                   "collective": ("one in-place all-reduce of the flat shared-gradient buffer per step "
                                                  f"({args.backend})") if n_ranks > 1 else None}
            if "rgb_lpips" in train:
                out["train_step_ms_lpips"] = train["rgb_lpips"]

            def train_traffic(prefix, mode_):
                """HBM bytes per launch of a backward kernel family from the committed PMC passes of the fitting step
                (profiles/traffic_train.json, written by profiles/train_pmc.sh), only if batch and mode match."""
                try:
                    t = json.load(open(os.path.join(ROOT, "profiles", "traffic_train.json")))
                    t = t.get("regimes", {}).get(mode_, t)       # {"regimes": {"3dmm": {...}, "rgb": {...}, "3dmm_tuned": {...}, ...}}
                    if t.get("batch") != train_B or t.get("mode") != mode_:
                        return None
                    for k, v in t.items():
                        if k.startswith(prefix):
                            return v["hbm_bytes"]
                except Exception:
                    pass
                return None

            def roofline_train(ev, step_ms, mode_="3dmm"):
                """Roofline objects of the BACKWARD kernel families of one fitting step ('train-step ms ... plus
                roofline fraction'), from HIP events around every launch of a second pass of the same step (`kernel_events`).
                  bwd-data GEMMs: algorithmic flops 2 M N K of the adjoint conv (gradients run in bf16x3: 3 MFMAs per product ->
                    2.5 PF / 3 = 833 TF ceiling); the figure includes the split-K reducer launches that follow a GEMM;
                  pointwise_bwd: bytes of the full-size tensors it reads + writes against HBM;
                  ray-march backward (both passes + the zero fill of d planes): against a floor of the re-gather through L2 (the
                    forward's 12-line gather per sample / 34.5 TB/s) + decoder recompute and adjoint (2 x the forward decoder flops
                    at the split-operand rate) + the sorted dL/dF stream once out and once back through HBM."""
                res_ = {"batch": train_B, "step_ms": step_ms}
                peak_g = MFMA_BF16_PEAK_TFLOPS / 3
                tot_ms = tot_fl = 0.0
                for key, name in (("bwd_data", "3x3 bwd-data (modconv_bf16_kernel, mode CONV3X3_BWD)"),
                                  ("bwd_data_up", "adjoint of the up-sampling conv (mode CONVS2_BWD, four parity phases merged in one kernel)"),
                                  ("bwd_data_1x1", "96-channel toRGB adjoint (mode CONV1X1)")):
                    if key in ev:
                        ms, fl, n = ev[key]
                        tot_ms, tot_fl = tot_ms + ms, tot_fl + fl
                        res_[key] = {"kernel": name, "ms_per_step": ms, "launches_per_step": n,
                                     "achieved": fl / (ms * 1e-3) / 1e12, "frac": fl / (ms * 1e-3) / 1e12 / peak_g,
                                     "traffic": train_traffic({"bwd_data": "modconv_bf16_kernel<2, 2, 9", "bwd_data_up": "modconv_bf16_kernel<2, 2, 0",
                                                               "bwd_data_1x1": "modconv_bf16_kernel<2, 2, 1"}[key], mode_)}
                if tot_ms > 0:
                    res_["bwd_data_gemms"] = {"bound": "mfma", "achieved": tot_fl / (tot_ms * 1e-3) / 1e12, "peak": peak_g,
                                              "unit": "TFLOP/s", "frac": tot_fl / (tot_ms * 1e-3) / 1e12 / peak_g,
                                              "ms_per_step": tot_ms, "share_of_step": tot_ms / step_ms}
                # generator being tuned: the weight-gradient GEMM family (wgrad_bf16_kernel<...> / wgrad_kernel<...> + their split-K
                # reducers; K = positions), algorithmic 2 * positions * Cin * Cout * taps against the same 833 TF (bf16x3)
                wg_ms = wg_fl = 0.0
                for key, name in (("wgrad", "3x3 weight gradient (wgrad_bf16_kernel<7,7> + wgrad_reduce_tiled_kernel)"),
                                  ("wgrad_up", "weight gradient of the up-sampling conv (parity images of the y_t gradient) + reducer"),
                                  ("wgrad_1x1", "96-channel toRGB weight gradient (wgrad_kernel<1,*>, exact fp32 MFMA) + reducer")):
                    if key in ev:
                        ms, fl, n = ev[key]
                        wg_ms, wg_fl = wg_ms + ms, wg_fl + fl
                        res_[key] = {"kernel": name, "ms_per_step": ms, "launches_per_step": n,
                                     "achieved": fl / (ms * 1e-3) / 1e12, "frac": fl / (ms * 1e-3) / 1e12 / peak_g,
                                     "traffic": train_traffic({"wgrad": "wgrad_bf16_kernel<7, 7", "wgrad_up": "wgrad_up_bf16_kernel",
                                                               "wgrad_1x1": "wgrad_kernel<1"}[key], mode_)}
                if wg_ms > 0:
                    res_["wgrad_gemms"] = {"bound": "mfma", "achieved": wg_fl / (wg_ms * 1e-3) / 1e12, "peak": peak_g,
                                           "unit": "TFLOP/s", "frac": wg_fl / (wg_ms * 1e-3) / 1e12 / peak_g,
                                           "ms_per_step": wg_ms, "share_of_step": wg_ms / step_ms,
                                           "algorithmic_gflop_per_step": wg_fl / 1e9}
                if "pointwise_bwd" in ev:
                    ms, by, n = ev["pointwise_bwd"]
                    gbs = by / (ms * 1e-3) / 1e9
                    res_["pointwise_bwd"] = {"bound": "hbm", "kernel": "pointwise_bwd_kernel (+ its partial-sum reducer)",
                                             "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                             "frac_of_measured_copy_ceiling_6290": gbs / 6290.0,
                                             "ms_per_step": ms, "launches_per_step": n, "share_of_step": ms / step_ms,
                                             "traffic": train_traffic("pointwise_bwd_kernel", mode_),
                                             "algorithmic_bytes_per_step": by}
                if "raymarch_bwd" in ev:
                    ms, fr, n = ev["raymarch_bwd"]
                    gb = fr * r * s_tot * 3 * 4 * 32 * 4
                    fl = fr * r * s_tot * 2.0 * (32 * 64 + 64 * 33)
                    # round 5: pass 2 is sort + gather (csrc/raymarch_rows.hip) — the re-gather through L2 and the decoder recompute +
                    # adjoint stay; the scatter is a stream: dL/dF (128 B) + (ix, wy) of every sample written to its sorted slot in
                    # two planes (the third is the mirror) and read back once, through HBM
                    stream = fr * r * s_tot * 2 * (128 + 8) * 2.0
                    floor = (gb / (L2_PEAK_GBS * 1e9) + 2.0 * fl / (MFMA_BF16_PEAK_TFLOPS / 3 * 1e12) + stream / (HBM_PEAK_GBS * 1e9)) * 1e3
                    # (generator tuned: the decoder-gradient pass raymarch_bwd_tiles_kernel IS the dL/dF producer, there is no df kernel)
                    producer = "raymarch_bwd_tiles_kernel" if mode_.endswith("_tuned") else "raymarch_bwd_df_kernel"
                    tr = [train_traffic(k, mode_) for k in (producer, "raymarch_bwd_rows_kernel", "raymarch_bwd_bins_kernel")]
                    res_["raymarch_bwd"] = {"bound": "l2+mfma+hbm",
                                            "kernel": "raymarch_kernel<GRADS> (compositing adjoint from the saved state) + sort (bins, scan, "
                                                      "place) + raymarch_bwd_df_kernel (dL/dF to sorted slots; tuned: the decoder-gradient "
                                                      "pass) + raymarch_bwd_rows_kernel (row tiles on the bf16 matrix pipe) + zero fill",
                                            "ms_per_step": ms, "ms_per_frame": ms / max(fr, 1), "floor_ms_per_step": floor,
                                            "frac": floor / ms, "share_of_step": ms / step_ms,
                                            "sorted_entries_per_frame": r * s_tot * 2, "stream_bytes_per_step": stream,
                                            "rows_scratch": dict(_rows_stats()),
                                            "traffic": sum(tr) if all(t is not None for t in tr) else None}
                if "raymarch" in ev:
                    res_["raymarch_fwd_ms_per_step"] = ev["raymarch"][0]
                fwd = sum(ev[k][0] for k in ev if k.startswith("modconv"))
                if fwd:
                    res_["conv_fwd_ms_per_step"] = fwd
                return res_
            out["roofline_train"] = {"rgb": roofline_train(trgb["kernel_events"], trgb["frozen"][train_B]["step_ms"], "rgb"),
                                     "3dmm": roofline_train(t3["kernel_events"], t3["frozen"][train_B]["step_ms"], "3dmm"),
                                     # the reference's regime after tune_iter (750 000 of its 800 000 default iterations,
                                     # train_rgb.py:132-134,162,193; trainer_rgb.py:69-71): all generator parameters get gradients
                                     "tuned": {"rgb": roofline_train(trgb["kernel_events_tuned"], trgb["tuned"][train_B]["step_ms"], "rgb_tuned"),
                                               "3dmm": roofline_train(t3["kernel_events_tuned"], t3["tuned"][train_B]["step_ms"], "3dmm_tuned")}}
            out["train_phases_ms_3dmm_generator_tuned"] = t3["tuned"][train_B]["phases_ms"]
        if fit is not None:
            out["fit_rgb"] = fit
        if fit3 is not None:
            out["fit_3dmm_sharded"] = fit3
        if audio is not None:
            out["audio_reenactment"] = audio
        if state is not None:
            t_cpu = time.perf_counter()
            def hip_image(ws_, c_, us_, ui_):
                gen.conv_precision, gen.sr_conv_precision, gen.sr_storage = cfg.conv_precision, cfg.sr_conv_precision, cfg.sr_storage
                with torch.no_grad():
                    return gen.synthesis(ws_.to(dev), c_.to(dev), noise_mode="const", u_strat=us_.to(dev), u_imp=ui_.to(dev))["image"].cpu()
            out["cpu_baseline"] = cpu_baseline(cfg, state, args.cpu_runs, args.cpu_warmup, args.cpu_n1, check=hip_image)
            leg_s["cpu_baseline"] = round(time.perf_counter() - t_cpu, 2)
        out["leg_seconds"] = leg_s
        write_detail(out, args.detail)
        line = json.dumps(compact_line(out))
        assert len(line) <= LINE_BYTE_BUDGET, len(line)
        sys.stdout.flush()
        sys.stderr.flush()
        os.dup2(stdout_fd, 1)
        print(line, flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""bench.py — rendered 512^2 frames/s at 96 depth samples (BASELINE.json metric, config 2).

One "step" = one pass of the hot path over one batch of synthetic input:
``generator.synthesis(ws[B,14,512], c[B,25], noise_mode='const')`` for the
``ffhq512_128`` preset (random-init EG3D weights, random latents + gaussian cameras,
explicit sampling uniforms), forward only, inputs resident in HBM before the
timed region.  N>1: one process per GPU, frames are independent → weak scaling,
no data-path collective; only the timing is reduced (MAX).

Prints ONE JSON line on rank 0 (see the contract in the task description).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32 dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_bf16 dense peak
SPLIT_MFMAS = {"f16x3": 3, "bf16x3": 3, "bf16x6": 6}   # 16-bit MFMAs per algorithmic (fp32) product on the split paths
SPLIT_ELEM = {"f16x3": "f16", "bf16x3": "bf16", "bf16x6": "bf16"}
MFMA_F16_PEAK_TFLOPS = 2500.0   # v_mfma_f32_32x32x16_f16 dense peak (same rate as bf16)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="frames per step per GPU")
    ap.add_argument("--preset", default="ffhq512_128")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the fitting-step leg (train_step_ms)")
    ap.add_argument("--train-batch", type=int, default=2, help="frames per fitting step per GPU")
    ap.add_argument("--train-steps", type=int, default=6)
    ap.add_argument("--no-sweep", action="store_true",
                    help="skip the batch-size sweeps (render B = 1, 4, 16; fitting step B = 1, 4; SURVEY.md section 8d)")
    ap.add_argument("--cpu-runs", type=int, default=2)
    ap.add_argument("--precision", default=None, choices=["fp32", "f16x3", "bf16x3", "bf16x6"],
                    help="conv GEMM arithmetic of the headline leg (default: the preset's conv_precision)")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the second render leg on the exact fp32 kernel")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend (nccl = RCCL; gloo + --share-device: developer check of the N > 1 code "
                         "path on a 1-GPU box, numbers not comparable)")
    ap.add_argument("--share-device", action="store_true", help="developer: every rank uses cuda:0")
    ap.add_argument("--no-f16-leg", action="store_true",
                    help="skip the legs with the super-resolution / all convs on the single-pass fp16 MFMA path")
    return ap.parse_args()


def cpu_baseline(cfg, state, runs: int):
    """Oracle (kind 'port') timed on this box's host cores: B=1 synthesis, same workload."""
    from oracle import eg3d_oracle as O
    from tests.util import make_inputs
    ws, c, us, ui = make_inputs(cfg, 1, seed=10)
    with torch.no_grad():
        O.synthesis(state, cfg, ws, c, us, ui)          # warm-up
        t = time.perf_counter()
        for _ in range(runs):
            O.synthesis(state, cfg, ws, c, us, ui)
        dt = (time.perf_counter() - t) / runs
    return {"value": 1.0 / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{runs} x synthesis(B=1) of the {cfg.name} workload with the fp32 PyTorch-CPU oracle, "
                      f"{dt:.2f} s/frame"}


def _pct(ms_list):
    """median / p10 / p90 of a list of per-step milliseconds (HIP event pairs)."""
    v = sorted(ms_list)
    n = len(v)
    if n == 0:
        return None
    q = lambda f: v[min(n - 1, max(0, int(round(f * (n - 1)))))]
    return {"median": q(0.5), "p10": q(0.1), "p90": q(0.9), "n": n}


class _FitArgs:
    """Flags of code/train_3dmm.py that reach the step."""
    out_pose = False
    person_2 = False
    params_len = 76
    size = 256
    batch_size = 1
    lr = 3e-4
    latent_dim_style = 512
    latent_dim_shape = 50
    generator_seed = 0


def train_leg(args, cfg_name, dev, rank, world, dist):
    """BASELINE config 3/4 mechanics: 3DMM-driven latent-basis fitting, generator frozen, L2 loss at 256^2
    (LPIPS weights are not available offline), Adam 3e-4; frames sharded over ranks, ONE flattened
    all-reduce of the shared gradients per step.  Returns max-over-ranks ms per step."""
    from hfa_gp_amd.trainer import Trainer
    from tests.util import look_at_label
    fa = _FitArgs()
    fa.generator_preset = cfg_name
    torch.manual_seed(0)
    tr = Trainer(fa, dev, rank=rank, world_size=world, mode="3dmm")
    g = torch.Generator().manual_seed(40 + rank)

    def inputs(B):
        real = (0.5 * torch.randn(B, 3, fa.size, fa.size, generator=g)).clamp(-1, 1).to(dev)
        params = torch.randn(B, fa.params_len, generator=g).to(dev)
        label = look_at_label(math.pi / 2 + 0.3 * torch.randn(B, generator=g),
                              math.pi / 2 + 0.155 * torch.randn(B, generator=g), flipped=False).to(dev)
        return real, params, label

    def timed(steps, B):
        """(max-over-ranks ms per step, {phase: mean ms} from HIP events on the launch stream)."""
        real, params, label = inputs(B)
        for _ in range(2):
            tr.gen_update(real, label.clone(), params)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        tr.timing = {}
        t0 = time.perf_counter()
        for _ in range(steps):
            l2, _, _ = tr.gen_update(real, label.clone(), params)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        spans, tr.timing = tr.timing, None
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        assert torch.isfinite(l2)
        phases = {k: sum(a.elapsed_time(b) for a, b in v) / max(len(v), 1) for k, v in spans.items()}
        return dt / steps * 1e3, phases

    B = args.train_batch
    out = {"frozen": {}, "tuned": {}}
    batches = [B] if args.no_sweep else sorted({1, B, 4})
    for b in batches:
        ms, phases = timed(args.train_steps, b)
        out["frozen"][b] = {"step_ms": ms, "phases_ms": phases}
    # after tune_iter the reference also trains the generator (trainer_rgb.py:69-71): all 30.7 M parameters get
    # gradients and are all-reduced with the basis / driver gradients
    tr.tune_generator()
    ms, phases = timed(max(2, args.train_steps // 2), B)
    out["tuned"][B] = {"step_ms": ms, "phases_ms": phases}
    return out, B


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); no CPU fallback")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from hfa_gp_amd.config import PRESETS
    from hfa_gp_amd.generator import TriPlaneGenerator
    from tests.util import make_inputs, state_cpu

    cfg = PRESETS[args.preset]()
    gen = TriPlaneGenerator(cfg, seed=0).requires_grad_(False)
    state = state_cpu(gen) if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    gen = gen.to(dev)
    B = args.batch
    # each rank renders its own frames (frame-parallel): different seed per rank
    ws, c, us, ui = make_inputs(cfg, B, seed=10 + rank)
    ws, c, us, ui = ws.to(dev), c.to(dev), us.to(dev), ui.to(dev)

    def step():
        return gen.synthesis(ws, c, noise_mode="const", u_strat=us, u_imp=ui)["image"]

    def render_leg(precision, sr_precision=None):
        """W warm-up + K timed steps with the conv GEMMs in ``precision`` (super-resolution blocks: ``sr_precision``
        when given); (seconds max over ranks, event table)."""
        gen.conv_precision = precision
        gen.sr_conv_precision = sr_precision
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        gen.timing = {}
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(args.steps):
            img = step()
            marks[i + 1].record()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        timing, gen.timing = gen.timing, None
        timing["step"] = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
        if dist is not None:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        assert torch.isfinite(img).all()
        return dt, timing

    def sweep_leg(b, precision, steps=20, warmup=3):
        """batch sizes 1, 4 (and 16) beside the headline's 8; per-step HIP event pairs."""
        gen.conv_precision, gen.sr_conv_precision = precision, None
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

    prec = args.precision or cfg.conv_precision
    # the headline leg first, exactly as the contract words it (W warm-up steps, then K timed steps); the secondary
    # legs (other precisions, batch sweep) follow
    dt, timing = render_leg(prec)
    dt32 = timing32 = dtb3 = None
    if prec != "fp32" and not args.no_fp32_leg:
        dt32, timing32 = render_leg("fp32")
        if prec != "bf16x3":
            dtb3, _ = render_leg("bf16x3")
    dt16sr = dt16 = timing16 = None
    if not args.no_f16_leg:
        # the reference's CUDA defaults: fp32-class backbone, fp16 super-resolution; then every conv in fp16
        dt16sr, timing16 = render_leg(prec, "f16")
        dt16, _ = render_leg("f16")
    sweep = None
    if not args.no_sweep:
        sweep = {str(b): sweep_leg(b, prec) for b in (1, 4, 16) if b != B}
    gen.conv_precision, gen.sr_conv_precision = prec, None

    def agg(key, table=None):
        evs = (timing if table is None else table).get(key, [])
        ms = sum(e0.elapsed_time(e1) for e0, e1, _ in evs)
        units = sum(u for _, _, u in evs)
        return ms, units, len(evs)

    train = train_B = None
    if not args.no_train:
        torch.cuda.empty_cache()
        train, train_B = train_leg(args, args.preset, dev, rank, world, dist)

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
        frames = world * B * args.steps
        rm_ms, rm_bytes, rm_n = agg("raymarch")
        rm_gbs = rm_bytes / (rm_ms * 1e-3) / 1e9

        def f32_roofline(table):
            ms, flops, n = agg("modconv", table)
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
            ms, flops, n = agg("modconv_split")
            tf = flops / (ms * 1e-3) / 1e12
            peak = MFMA_BF16_PEAK_TFLOPS / SPLIT_MFMAS[prec]
            kd = {"f16x3": 4, "bf16x3": 2, "bf16x6": 3}[prec]
            roof = {"bound": "mfma", "kernel": f"modconv_bf16_kernel<{kd}> / upconv_bf16_kernel<{kd}> ({SPLIT_MFMAS[prec]} x "
                                               f"v_mfma_f32_32x32x16_{SPLIT_ELEM[prec]} per fp32 product)",
                    "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak,
                    "traffic": profiled_traffic(f"modconv_bf16_kernel<{kd}, 2, 9>"), "avg_launch_ms": ms / max(n, 1),
                    "launches": n, "mfma_16bit_tflops": tf * SPLIT_MFMAS[prec]}
        out = {
            "metric": "rendered 512^2 frames/sec (96 depth samples), whole job",
            "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if prec == "fp32" else f"f32 tensors and accumulation; conv GEMM products as {prec} "
                                                  f"(operands split into {SPLIT_ELEM[prec]} parts, {SPLIT_MFMAS[prec]} "
                                                  f"MFMAs per product" + (": 22 mantissa bits, fp32-class results)"
                                                                          if prec == "f16x3" else ")"),
            "data": "synthetic",
            "config": {"workload": f"{cfg.name}: synthesis(ws[B,14,512], c[B,25]) forward, 512x512 out, 128^2 rays x "
                                   f"(48+48) samples, random-init weights, random latents+cameras",
                       "frames_per_step_per_gpu": B, "parallelism": f"frame-parallel x{world}"},
            # dominant kernel by time: the modulated-conv implicit GEMM
            "roofline": roof,
            # the kernel north_star sets the HBM target on
            "roofline_raymarch": {"bound": "hbm", "kernel": "raymarch_kernel<3,3>", "achieved": rm_gbs,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rm_gbs / HBM_PEAK_GBS,
                                  "traffic": profiled_traffic("raymarch_kernel"),
                                  "avg_launch_ms": rm_ms / max(rm_n, 1), "launches": rm_n,
                                  "algorithmic_bytes_per_launch": rm_bytes / max(rm_n, 1)},
        }
        out["config"]["conv_precision"] = prec
        if dt16sr is not None:
            # not the headline: products of the super-resolution convs (value_f16_sr: the reference's own CUDA
            # precision split) or of all convs (value_f16) rounded to fp16, one fp16 MFMA per product
            out["value_f16_sr"] = frames / dt16sr
            out["value_f16"] = frames / dt16
            ms, flops, n = agg("modconv_f16", timing16)
            tf = flops / (ms * 1e-3) / 1e12
            out["roofline_f16_sr"] = {"bound": "mfma", "kernel": "modconv_bf16_kernel<1> / upconv_bf16_kernel<1> "
                                                                 "(1 x v_mfma_f32_32x32x16_f16 per product)",
                                      "achieved": tf, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": tf / MFMA_F16_PEAK_TFLOPS, "traffic": None,
                                      "avg_launch_ms": ms / max(n, 1), "launches": n}
        if dtb3 is not None:
            out["value_bf16x3"] = frames / dtb3      # bf16 hi+lo parts: ~5 % faster, product error 2^-16 instead of 2^-22
        if dt32 is not None:
            out["value_fp32_exact"] = frames / dt32
            out["roofline_fp32_exact"] = f32_roofline(timing32)
        out["step_ms"] = _pct(timing["step"])       # per-step HIP event pairs of the timed region (this rank)
        if sweep is not None:
            out["batch_sweep"] = sweep
        if train is not None:
            out["train_step_ms"] = train["frozen"][train_B]["step_ms"]
            out["train_step_ms_generator_tuned"] = train["tuned"][train_B]["step_ms"]
            # fwd / bwd / all-reduce / Adam from HIP events on the launch stream (mean ms per step, this rank)
            out["train_phases_ms"] = train["frozen"][train_B]["phases_ms"]
            out["train_phases_ms_generator_tuned"] = train["tuned"][train_B]["phases_ms"]
            out["train_step_ms_by_batch"] = {str(b): v["step_ms"] for b, v in train["frozen"].items()}
            out["train_config"] = {"workload": "3DMM-driven latent-basis fitting step (fwd + bwd + Adam), K=50, "
                                               "generator frozen, L2 at 256^2, synthetic frames",
                                   "frames_per_step_per_gpu": train_B,
                                   "collective": "one flattened all-reduce of shared grads per step" if world > 1 else None}
        if state is not None:
            out["cpu_baseline"] = cpu_baseline(cfg, state, args.cpu_runs)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""bench.py's launcher path (no GPU needed): `--gpus N` without a launcher must start N ranks itself, and a mismatch
between --gpus and the launcher's world size must be refused instead of silently running one rank."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_launcher_command_is_the_drivers_form():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "3"], 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")


def test_gpus_n_without_launcher_spawns_n_ranks():
    """On this GPU-less box every spawned rank stops at the 'needs an MI355X' check.  `--gpus 2` must have started the ranks
    under torch.distributed.run with a world size of 2 instead of running a single rank as round 1 did: the parent says so
    before it launches, and every rank that got as far as the check reports 'rank r of 2'.  (torchrun tears the other rank
    down as soon as the first one exits, so only ONE such line is guaranteed — asserting two was a race: VERDICT r3 #12.)"""
    import re
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=_env(), capture_output=True, text=True, timeout=600)
    assert run.returncode != 0
    assert "starting 2 ranks under torch.distributed.run" in run.stderr, run.stderr[-2000:]
    ranks = re.findall(r"bench.py needs an MI355X .*\[rank (\d) of (\d)\]", run.stderr)
    assert 1 <= len(ranks) <= 2 and all(w == "2" for _, w in ranks), run.stderr[-2000:]


def test_world_size_mismatch_is_refused():
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"],
                         env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=600)
    assert run.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in run.stderr


def test_rccl_choice_parser(tmp_path):
    """bench.py `allreduce_us.algo`: RCCL's TUNING lines -> {payload bytes: algorithm, protocol}; a one-rank log has none and says why."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    log = tmp_path / "rccl.log"
    log.write_text("h:1:2 [0] NCCL INFO RCCL version 2.26.6-HEAD\n"
                   "h:1:2 [0] NCCL INFO AllReduce: 1464320 Bytes -> Algo 1 proto 0 time 23.5\n"
                   "h:1:2 [0] NCCL INFO AllReduce: 96468992 Bytes -> Algo 0 proto 2 time 1440.0\n")
    got = bench.rccl_choices(str(log), 8)
    assert got["by_payload_bytes"]["1464320"] == {"algo": "Ring", "proto": "LL", "model_time_us": 23.5}
    assert got["by_payload_bytes"]["96468992"]["algo"] == "Tree" and got["by_payload_bytes"]["96468992"]["proto"] == "Simple"
    log.write_text("h:1:2 [0] NCCL INFO RCCL version 2.26.6-HEAD\n")
    assert "one rank" in bench.rccl_choices(str(log), 1)["note"]
    assert "no RCCL log" in bench.rccl_choices(str(tmp_path / "absent.log"), 2)["note"]


def _bench_module():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_the_printed_line_stays_parseable_and_short():
    """round 5's line had grown to 21.6 KB and the driver recorded `parsed: null`.  `compact_line` turns the full result
    object of a real run (canned: profiles/r05_bench_line.json, the 21.6 KB one) into the line that is printed: valid JSON, at most
    LINE_BYTE_BUDGET (6000) bytes, every contract key present, `roofline` and `cpu_baseline` with their required members."""
    import json
    bench = _bench_module()
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_line.json")))
    assert len(json.dumps(full)) > 20000
    text = json.dumps(bench.compact_line(full))
    assert len(text) <= bench.LINE_BYTE_BUDGET == 6000, len(text)
    line = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert abs(line["value"] - full["value"]) < 1e-3 * full["value"] and line["steps"] == 20 and line["warmup"] == 5
    assert set(line["roofline"]) == {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches"}
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["train_step_ms_3dmm_generator_tuned"] > 0 and line["roofline_train_frac_ms"]["tuned_3dmm"]["wgrad_gemms"][0] > 0
    # an 8-rank line carries the per-rank rates and still fits; a pathological one sheds its optional groups instead of growing
    full8 = dict(full, n_gpus=8, per_rank_frames_per_s=[830.0 + i for i in range(8)], per_rank_spread=0.01)
    assert len(json.dumps(bench.compact_line(full8))) <= 6000
    huge = dict(full, leg_seconds={f"leg{i}": 1.0 for i in range(600)})
    small = bench.compact_line(huge)
    assert len(json.dumps(small)) <= 6000 and "leg_seconds" not in small and "roofline" in small

"""bench.py's launcher path (no GPU needed): `--gpus N` without a launcher must start N ranks itself, and a mismatch
between --gpus and the launcher's world size must be refused instead of silently running one rank."""
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

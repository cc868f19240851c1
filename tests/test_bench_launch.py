"""bench.py's N>1 entry point on a box without a GPU (`--dry-run`: gloo, no kernels): `--gpus N`
must end up as N ranks, whether bench.py launches them itself or a launcher did."""
import json
import os
import socket
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def _last_json(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_gpus_flag_launches_that_many_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1
    assert j["frames_per_step_all_ranks"] == 1000 + 1001          # both ranks counted by the ONE all-reduce
    assert "all_reduce" in j["config"]["collective"] and j["config"]["parallelism"].endswith("dp2")


def test_single_rank_default():
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "0", "--dry-run"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 1 and j["config"]["collective"] == "none"


def test_under_a_launcher_as_the_driver_invokes_it():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH,
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _last_json(r.stdout)["n_gpus"] == 2


def test_rank_count_must_match_the_flag():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--dry-run"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "--gpus 4" in (r.stderr + r.stdout)


def test_eight_ranks_as_the_scaling_run_will_start_them():
    """--gpus 8 (the driver's widest scaling point): eight ranks, every one counted once by the ONE all-reduce, per-rank
    times and frame counts in the line (imbalance is the only thing that can cost the scaling)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, env=_env(), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 8 and j["config"]["parallelism"].endswith("dp8")
    assert j["frames_per_step_all_ranks"] == sum(1000 + k for k in range(8))
    assert j["per_rank"]["frames"] == [1000 + k for k in range(8)] and len(j["per_rank"]["ms_per_step"]) == 8

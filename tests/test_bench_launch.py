"""bench.py's N>1 entry point on a box without a GPU (`--dry-run`: gloo, no kernels): `--gpus N`
must end up as N ranks, whether bench.py launches them itself or a launcher did, and the run must go through the
PRODUCT'S sharding - the global minibatch built on every rank, `parallel.shard_batch`, `parallel.ShardedChainLoss` with
its one all-reduce - with a stand-in for the per-rank kernels (BASELINE.json C5)."""
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
    _check_sharded(j, 2)
    assert "all_reduce" in j["config"]["collective"] and j["config"]["parallelism"].endswith("dp2")


def _check_sharded(j, world):
    """what the sharded path can get wrong: an utterance lost or owned twice, frames that do not add up (both ranks
    counted by the ONE all-reduce), a global loss that is not the single-process loss, an unbalanced deal"""
    sh = j["sharding"]
    assert "shard_batch" in sh["partitioner"] and "ShardedChainLoss" in sh["partitioner"]
    assert sh["every_utterance_owned_once"] and sh["frames_add_up"] and sh["loss_equals_single_process"]
    assert j["frames_per_step_all_ranks"] == j["global_frames"] == sum(j["per_rank"]["frames"])
    assert abs(j["loss"] - j["loss_single_process"]) <= 1e-5 * abs(j["loss_single_process"])
    assert len(j["per_rank"]["ms_per_step"]) == world and len(j["per_rank"]["longest_sequence"]) == world
    assert j["config"]["global_batch"] == sum(j["per_rank"]["utterances"])
    assert sh["frames_imbalance_max_over_mean"] < 1.15            # (tiny shapes; the C3 deal is within 5 %: test_parallel.py)


def test_single_rank_default():
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "0", "--dry-run"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 1 and j["config"]["collective"] == "none"
    _check_sharded(j, 1)


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
    """--gpus 8 (the driver's widest scaling point, BASELINE.json's C5): eight ranks over the product's partitioner, every
    utterance owned once, per-rank times / frames / longest sequences in the line (imbalance is the only thing that can cost
    the scaling)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, env=_env(), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 8 and j["config"]["parallelism"].endswith("dp8")
    _check_sharded(j, 8)
    assert "C5: global B=" in j["config"]["workload"] and "batch-sharded over 8 GPUs" in j["config"]["workload"]

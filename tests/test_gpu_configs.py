"""BASELINE.json configs the parity suite did not reach at full size: C4 (B=32, T=2000, 8408 pdfs - the
"HBM-roofline" configuration: 4 time segments, gate kernels, the wide-row recursion and the one-frame
occupancy kernel at scale), the training example run end to end, half-precision network outputs
against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle as orc
from helpers import rel_err
from pychain_amd import ChainFunction, ChainGraphBatch, ChainLoss, synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c4_full_size_properties():
    """C4 at full size through size-independent properties (the oracle would need minutes): every live
    frame's occupancies sum to one, ok, bit-reproducible run to run, and the same gradient with and
    without the overlapped schedule."""
    from pychain_amd import _lib
    w = syn.make_workload("C4", device=DEV)
    B, T = w["cfg"]["B"], w["cfg"]["T"]
    gb = ChainGraphBatch(w["den_graph"], B)
    grads, objfs = [], []
    for rep in range(2):
        xx = w["x"].clone().requires_grad_(True)
        o = ChainFunction.apply(xx, w["lengths"], gb, 1e-5)
        o.backward()
        assert int(ChainFunction.last_bad_count.sum()) == 0
        grads.append(xx.grad)
        objfs.append(float(o.detach()))
    assert torch.equal(grads[0], grads[1]) and objfs[0] == objfs[1] and np.isfinite(objfs[0])
    rows = grads[0].sum(-1)
    assert torch.allclose(rows, torch.ones_like(rows), atol=3e-4)
    assert float(grads[0].min()) >= 0.0
    del grads[1]
    with _lib.option("den_segments", 1):
        xx = w["x"].clone().requires_grad_(True)
        ChainFunction.apply(xx, w["lengths"], gb, 1e-5).backward()
    assert torch.equal(xx.grad, grads[0])


def test_c4_full_length_slice_vs_oracle():
    """B = 3 at C4's full T = 2000 and D = 8408 (one sequence shorter): four time segments, the gated
    schedule, the recursion form for rows wider than 16 KiB and the one-frame occupancy kernel, against
    the fp32 restatement of the reference."""
    cfg = syn.CONFIGS["C4"]
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    L = torch.tensor([2000, 2000, 1377])
    x = syn.make_input(3, 2000, cfg["D"], seed=77)
    xx = x.to(DEV).requires_grad_(True)
    o = ChainFunction.apply(xx, L, ChainGraphBatch(den, 3), 1e-5)
    o.backward()
    assert int(ChainFunction.last_bad_count.sum()) == 0
    ro, rg = orc.chain_function(x, L, ChainGraphBatch(den, 3), 1e-5)
    assert abs(float(o.detach()) - ro) <= 1e-4 * abs(ro)
    assert rel_err(xx.grad.cpu().numpy(), rg) <= 1e-4
    assert bool((xx.grad[2, 1377:] == 0).all())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_half_precision_network_output_vs_oracle(dtype):
    """fp16 / bf16 nnet_output (an autocast training loop): the HIP path evaluates the ROUNDED values in
    fp32 - so does the oracle here, fed the same rounded values - and returns the gradient in the
    input's dtype: agreement to the rounding of that dtype, loss to 1e-4."""
    w = syn.make_workload("C1")
    xh = w["x"].to(dtype)                                           # the rounding happens here, once
    xx = xh.to(DEV).requires_grad_(True)
    loss = ChainLoss(w["den_graph"], 1e-5)(xx, w["lengths"], w["num_graphs"])
    loss.backward()
    assert xx.grad.dtype == dtype and int(ChainFunction.last_bad_count.sum()) == 0
    ref_l, ref_g = orc.chain_loss(xh.float(), w["lengths"], w["den_graph"], w["num_graphs"], 1e-5, avg=True)
    assert abs(float(loss.detach()) - float(ref_l)) <= 1e-4 * abs(float(ref_l))
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11      # half an ulp of the returned dtype, relative
    assert rel_err(xx.grad.float().cpu().numpy(), ref_g) <= eps + 1e-4


def test_training_example_runs_and_the_loss_falls():
    """examples/train_tdnn.py (pychain_example-style loop: TDNN -> ChainLoss -> AdamW through the
    `pychain` alias package) for 8 steps on one GPU; the script asserts that the loss went down."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "examples", "train_tdnn.py"), "--steps", "8",
                        "--batch", "8", "--frames", "120"], capture_output=True, text=True, timeout=600,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "loss" in r.stdout and "-> " in r.stdout


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the driver's multi-GPU box)")
def test_training_example_two_ranks_over_rccl():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(REPO, "examples", "train_tdnn.py"), "--steps", "8", "--batch", "8",
                        "--frames", "120"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "on 2 GPU(s)" in r.stdout


def _launch_one_rank(script_args, timeout=900, env_extra=None):
    """`python -m torch.distributed.run --nproc-per-node 1 <script>`: one rank under the launcher, RCCL's environment exported."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.pop("GPU_MAX_HW_QUEUES", None)                 # (the scripts set their own default)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                           "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_args,
                          capture_output=True, text=True, timeout=timeout, env=env)


def test_training_example_one_rank_over_rccl():
    """RCCL executed for real on the one GPU of the test box: examples/train_tdnn.py under the launcher with
    ONE rank, backend "nccl" (= RCCL), `--force-collective` - RCCL's initialisation, DDP's bucket all-reduce of the parameter
    gradients and ShardedChainLoss's all-reduce of the loss call's kernel-written device totals beside the loss's side streams
    and spin-wait kernels.  The script exits non-zero if a kernel of the loss gives up into `bad` or the loss does not fall."""
    r = _launch_one_rank([os.path.join(REPO, "examples", "train_tdnn.py"), "--steps", "8", "--batch", "8", "--frames", "120",
                          "--force-collective"])
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "collective backend nccl, world 1 (forced)" in r.stdout and "on 1 GPU(s)" in r.stdout, r.stdout[-800:]


def test_bench_one_rank_over_rccl_beside_the_plain_step():
    """`bench.py --gpus 1` as the driver launches it for N > 1 - under torch.distributed.run - with the step's all-reduce forced
    through RCCL, against the plain one-process run: same batch, same frames, `bad` = 0, and the forced collective costs the C3
    step less than 5 %.  (It cost 37 % until bench.py asked for more hardware queues than the runtime's default of four: RCCL's
    stream shared a queue with one of the loss's three - profiles/r06_rccl_one_rank.txt; the last leg shows that it still does
    with GPU_MAX_HW_QUEUES=4, i.e. what the variable is for.)"""
    import json
    common = ["--gpus", "1", "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-other-workloads", "--no-rooflines",
              "--no-fresh-num-graphs"]
    def both():
        r = _launch_one_rank([os.path.join(REPO, "bench.py")] + common + ["--force-collective"])
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        forced = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + common, capture_output=True, text=True, timeout=900,
                           env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
        plain = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        assert "forced" in forced["config"]["collective"] and plain["config"]["collective"] == "none"
        assert forced["n_bad"] == 0 and plain["n_bad"] == 0
        assert forced["sharding"]["frames_add_up"] and forced["per_rank"]["frames"] == plain["per_rank"]["frames"]
        return forced, plain
    from helpers import record_parity
    # (two processes of ten steps each, one after the other on a box that has just run the rest of the suite: a pair that misses the
    # bound is measured again - the bound is on what the collective costs, not on the box's moment)
    for attempt in range(3):
        forced, plain = both()
        record_parity("bench_one_rank_rccl", forced_ms=forced["ms_per_step"], plain_ms=plain["ms_per_step"],
                      ratio=forced["ms_per_step"] / plain["ms_per_step"], bound=1.05, attempt=attempt)
        if forced["ms_per_step"] <= 1.05 * plain["ms_per_step"]:
            break
    assert forced["ms_per_step"] <= 1.05 * plain["ms_per_step"], (forced["ms_per_step"], plain["ms_per_step"])
    four = _launch_one_rank([os.path.join(REPO, "bench.py")] + common + ["--force-collective"], env_extra={"GPU_MAX_HW_QUEUES": "4"})
    assert four.returncode == 0, four.stdout[-1500:] + four.stderr[-1500:]
    four = json.loads([l for l in four.stdout.splitlines() if l.startswith("{")][-1])
    record_parity("bench_one_rank_rccl_four_hw_queues", forced_ms=four["ms_per_step"], plain_ms=plain["ms_per_step"],
                  ratio=four["ms_per_step"] / plain["ms_per_step"])
    assert four["n_bad"] == 0                                      # slower (a shared queue), never wrong


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="needs a box with ONE visible GPU")
def test_bench_with_more_ranks_than_devices_fails_fast():
    """`bench.py --gpus 2` on a one-GPU box: one clear line, at once - not two ranks dying in RCCL's initialisation."""
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode != 0 and "only 1 HIP device" in (r.stderr + r.stdout), (r.stdout[-300:], r.stderr[-300:])
    assert time.time() - t0 < 120


def test_training_example_two_ranks_share_one_gpu_over_gloo():
    """The sharded step in TWO processes on device tensors where only one GPU is there: both ranks on cuda:0, collectives
    over gloo (they copy device tensors through the host) - the HIP kernels of two processes beside each other, DDP's bucket
    all-reduce, ShardedChainLoss's one all-reduce of the loss call's device totals.  (The RCCL flavour of the same run needs
    two GPUs: the test above, skipped here.)"""
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(REPO, "examples", "train_tdnn.py"), "--steps", "8", "--batch", "8",
                        "--frames", "120", "--backend", "gloo"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "on 2 GPU(s)" in r.stdout
    import re
    ends = re.findall(r"rank (\d) of 2: global loss ([-\d.]+) -> ([-\d.]+)", r.stdout)
    assert len(ends) == 2 and ends[0][1:] == ends[1][1:], r.stdout[-600:]      # both ranks report the same global loss


def test_bench_two_ranks_share_one_gpu():
    """`bench.py --gpus 2 --share-gpu`: the bench's sharded path with the REAL kernels in two processes (both on this box's one
    GPU, collectives over gloo): every utterance of the global minibatch (128) owned once, frames add up, no `bad`, one line from
    rank 0 with `n_gpus` = 2 and the per-rank view - what the driver's scaling run exercises with one GPU per rank over RCCL."""
    import json
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--share-gpu", "--steps", "3", "--warmup", "1",
                        "--no-grad-slab", "--no-rooflines"], capture_output=True, text=True, timeout=900,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-800:]
    d = json.loads(line[0])
    assert d["n_gpus"] == 2 and d["n_bad"] == 0 and d["value"] > 0 and d["config"]["global_batch"] == 128
    assert d["sharding"]["every_utterance_owned_once"] and d["sharding"]["frames_add_up"]
    assert len(d["per_rank"]["frames"]) == 2 and d["per_rank"]["utterances"] == [64, 64]

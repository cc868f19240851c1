"""N>1 path on CPU: world_size-2 gloo group exercising the sharding + single all-reduce
(the same code bench.py runs over RCCL).  The loss itself is replaced by the CPU oracle
here - the collective logic is what is under test."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pychain_amd import parallel, synthetic as syn


def test_shard_indices_balanced_and_sorted():
    L = syn.make_lengths(64, 1500, "ragged", seed=2)
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(0))
    L = L[perm]
    shards = [parallel.shard_indices(L, 8, r) for r in range(8)]
    allidx = torch.cat(shards).sort().values
    assert torch.equal(allidx, torch.arange(64))
    frames = [int(L[s].sum()) for s in shards]
    assert (max(frames) - min(frames)) / max(frames) < 0.05
    for s in shards:
        ls = L[s]
        assert bool((ls[:-1] >= ls[1:]).all())


from helpers import OracleChainLoss as _OracleLoss  # noqa: E402  (ChainLoss(avg=False) by the CPU oracle, with autograd)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = syn.make_workload("C1")
        L = torch.tensor([50, 37, 44, 50])
        x = syn.make_input(4, 50, 40, seed=5)
        numg = syn.make_num_graphs(L.tolist(), 40, seed=100, max_states=12)
        xs, ls, gs, idx = parallel.shard_batch(x, L, numg, world, rank)
        xs = xs.clone().requires_grad_(True)
        loss = parallel.ShardedChainLoss(w["den_graph"], 1e-5, avg=True, loss_cls=_OracleLoss)(xs, ls, gs)
        loss.backward()
        # the optional fused collective: scalars + whole-batch gradient slab in ONE all-reduce
        st, slab = parallel.allreduce_grad_slab(xs.grad, idx, 4, stats=torch.tensor([1.0, float(rank)]))
        out[rank] = (float(loss), idx.tolist(), xs.grad.numpy(), st.numpy().copy(), slab.numpy().copy())
    finally:
        dist.destroy_process_group()


def test_sharded_loss_matches_single_process():
    import oracle as orc
    world, port = 2, 29731 + os.getpid() % 1000
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    w = syn.make_workload("C1")
    L = torch.tensor([50, 37, 44, 50])
    x = syn.make_input(4, 50, 40, seed=5)
    numg = syn.make_num_graphs(L.tolist(), 40, seed=100, max_states=12)
    ref_loss, ref_grad = orc.chain_loss(x, L, w["den_graph"], numg, avg=True)
    for r in range(world):
        loss, idx, grad, st, slab = out[r]
        assert abs(loss - float(ref_loss)) <= 1e-5 * abs(float(ref_loss))
        np.testing.assert_allclose(grad, ref_grad[idx], atol=1e-6)
        # every rank holds the whole gradient after the optional slab all-reduce
        np.testing.assert_allclose(slab, ref_grad, atol=1e-6)
        np.testing.assert_allclose(st, [2.0, 1.0])


def test_training_example_two_ranks_ddp_on_gloo():
    """examples/train_tdnn.py as a launcher would start it, 2 ranks, no GPU: DistributedDataParallel around the model,
    ShardedChainLoss (ONE 3-float all-reduce per step), clip, optimizer step - the wiring of SURVEY.md §8(f)4 - with the
    per-rank loss evaluated by the CPU oracle stand-in.  The loss falls, and both ranks report the same GLOBAL loss."""
    import re
    import socket
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(repo, "tests"), os.path.join(repo, "oracle"), env.get("PYTHONPATH", "")])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(repo, "examples", "train_tdnn.py"),
           "--backend", "gloo", "--device", "cpu", "--loss-cls", "helpers:OracleChainLoss",
           "--steps", "8", "--batch", "3", "--frames", "40", "--pdfs", "40", "--states", "20", "--arcs", "60",
           "--hidden", "32", "--max-num-states", "8", "--lr", "0.01"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    seen = dict((int(m.group(1)), (float(m.group(2)), float(m.group(3))))
                for m in re.finditer(r"rank (\d) of 2: global loss (-?\d+\.\d+(?:e-?\d+)?) -> (-?\d+\.\d+(?:e-?\d+)?)", r.stdout))   # (the two ranks' lines may interleave)
    assert sorted(seen) == [0, 1], r.stdout
    assert seen[0] == seen[1]                         # the value every rank holds is the global one
    assert seen[0][1] < seen[0][0]                    # and it fell


def test_global_workload_is_the_same_on_every_rank_and_sharded_without_materialising_it():
    """bench.py --gpus N: every rank builds the same global minibatch (lengths, graphs), owns the utterances shard_batch deals
    to it and draws the network output of exactly those (synthetic.make_input_utterances) - the global [B,T,D] never exists."""
    world = 4
    g = syn.make_global_workload("C1", world)
    g2 = syn.make_global_workload("C1", world)
    assert torch.equal(g["lengths"], g2["lengths"]) and g["cfg"]["B_global"] == 2 * world
    assert torch.equal(g["num_graphs"].forward_transitions, g2["num_graphs"].forward_transitions)
    T, D = g["cfg"]["T"], g["cfg"]["D"]
    whole = syn.make_input_utterances(range(2 * world), T, D, seed=1)
    seen = []
    for r in range(world):
        xs, ls, gs, idx = parallel.shard_batch(None, g["lengths"], g["num_graphs"], world, r)
        assert xs is None and gs.batch_size == idx.numel() == 2
        assert torch.equal(ls, g["lengths"][idx])
        assert torch.equal(gs.forward_transitions, g["num_graphs"].forward_transitions[idx])
        assert torch.equal(syn.make_input_utterances(idx, T, D, seed=1), whole[idx])
        seen += idx.tolist()
    assert sorted(seen) == list(range(2 * world))

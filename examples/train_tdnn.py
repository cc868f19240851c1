#!/usr/bin/env python3
"""End-to-end LF-MMI training loop on MI355X in the style of pychain_example (reference
README.md:9), on synthetic data: a small TDNN -> ChainLoss -> AdamW, one process per GPU.

    python examples/train_tdnn.py --steps 30                                  # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \\
        examples/train_tdnn.py --steps 30                                     # 8 GPUs, RCCL over xGMI

`--backend gloo --device cpu --loss-cls module:Class` runs the same wiring (DistributedDataParallel around the
model, ShardedChainLoss, optimizer step) without a GPU, with the per-rank loss evaluated by a stand-in the CALLER
names: the HIP loss has no CPU form, and this example does not know one (tests/test_parallel.py passes its
oracle-backed test class).

Data parallelism as DESIGN.md §6 describes it: every rank owns its utterances, the loss kernels run
on the local shard only, DDP all-reduces the PARAMETER gradients, and the only LF-MMI-specific
collective is the 3-float all-reduce of ShardedChainLoss (global loss value / frame normaliser).
Code written against the reference imports unchanged (`from pychain.loss import ChainLoss`).
"""
import argparse
import os
import sys

# (more hardware queues than the runtime's default of four: the loss's three streams must not share one with RCCL's - a trainer
# that overlaps collectives with the loss loses a third of its step otherwise; bench.py, INTEGRATION.md "Streams and queues")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

import torch
import torch.distributed as dist
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pychain.graph import ChainGraphBatch  # noqa: E402  (alias of pychain_amd)
from pychain_amd import synthetic as syn  # noqa: E402
from pychain_amd.parallel import ShardedChainLoss  # noqa: E402


class TDNN(nn.Module):
    def __init__(self, feat_dim, hidden, num_pdfs, layers=4):
        super().__init__()
        blocks, d = [], feat_dim
        for i in range(layers):
            blocks += [nn.Conv1d(d, hidden, kernel_size=3, padding=2 ** min(i, 2), dilation=2 ** min(i, 2)),
                       nn.ReLU(), nn.BatchNorm1d(hidden)]
            d = hidden
        self.net = nn.Sequential(*blocks)
        self.out = nn.Conv1d(hidden, num_pdfs, kernel_size=1)

    def forward(self, feats):                       # [B,T,F] -> [B,T,D]
        return self.out(self.net(feats.transpose(1, 2))).transpose(1, 2).contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=16, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--pdfs", type=int, default=400)
    ap.add_argument("--states", type=int, default=300)
    ap.add_argument("--arcs", type=int, default=3000)
    ap.add_argument("--feat-dim", type=int, default=40)
    ap.add_argument("--lr", type=float, default=2e-3)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--max-num-states", type=int, default=80)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help='"nccl" is RCCL on ROCm')
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"])
    ap.add_argument("--force-collective", action="store_true",
                    help="ONE rank under a launcher (torch.distributed.run --nproc-per-node 1): initialise the backend, wrap the model in "
                         "DDP and run the loss's all-reduce although the world is one - RCCL beside the loss's side streams and "
                         "spin-wait kernels on a box with a single GPU")
    ap.add_argument("--loss-cls", default=None,
                    help="module:Class of a ChainLoss(den_graph, leaky, avg=False) stand-in for the per-rank loss")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if args.device == "cuda":
        # (fewer GPUs than ranks: only with gloo, whose collectives copy device tensors through the host - RCCL wants a GPU
        # per rank; the one-GPU test box runs two ranks on its GPU this way: tests/test_gpu_configs.py)
        gpu = local_rank if (args.backend == "nccl" or local_rank < torch.cuda.device_count()) else local_rank % torch.cuda.device_count()
        torch.cuda.set_device(gpu)
        dev = torch.device("cuda", gpu)
    else:
        dev = torch.device("cpu")
    use_dist = world > 1 or (args.force_collective and "WORLD_SIZE" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.device == "cuda" and args.backend == "nccl":
            dist.init_process_group(args.backend, device_id=dev)
        else:
            dist.init_process_group(args.backend)
    loss_cls = None
    if args.loss_cls:
        import importlib
        mod, _, attr = args.loss_cls.partition(":")
        loss_cls = getattr(importlib.import_module(mod), attr)

    torch.manual_seed(0)
    model = TDNN(args.feat_dim, args.hidden, args.pdfs).to(dev)
    if use_dist:
        model = nn.parallel.DistributedDataParallel(model, device_ids=[dev.index] if args.device == "cuda" else None)
    opt = torch.optim.AdamW(model.parameters(), lr=args.lr)
    den_graph = syn.make_den_graph(args.states, args.arcs, args.pdfs, seed=0)      # the shared "phone LM"
    criterion = ShardedChainLoss(den_graph, leaky_coefficient=1e-5, avg=True, loss_cls=loss_cls, force_collective=use_dist and world == 1)

    # a fixed synthetic training set per rank: features correlated with the numerator alignment
    gen = torch.Generator(device=dev).manual_seed(100 + rank)
    lengths = syn.make_lengths(args.batch, args.frames, "ragged", seed=7 + rank)
    num_graphs = syn.make_num_graphs(lengths.tolist(), args.pdfs, seed=500 + 1000 * rank, max_states=args.max_num_states)
    feats = torch.randn(args.batch, args.frames, args.feat_dim, generator=gen, device=dev)
    first = last = None
    for step in range(args.steps):
        opt.zero_grad(set_to_none=True)
        loss = criterion(model(feats), lengths, num_graphs)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        opt.step()
        if step == 0:
            first = float(loss)
        last = float(loss)
        if args.device == "cuda" and int(criterion.last_stats[2]) != 0:
            raise SystemExit("a kernel of the loss gave up or saw a bad value (bad count %d)" % int(criterion.last_stats[2]))
        if rank == 0 and (step % 5 == 0 or step == args.steps - 1):
            print("step %3d  LF-MMI loss per frame %.4f" % (step, last), flush=True)
    # the loss value is the GLOBAL one on every rank (ShardedChainLoss: one 3-float all-reduce per step)
    print("rank %d of %d: global loss %.6f -> %.6f" % (rank, world, first, last), flush=True)
    if rank == 0:
        print("loss %.4f -> %.4f over %d steps on %d %s" % (first, last, args.steps, world,
                                                             "GPU(s)" if args.device == "cuda" else "CPU rank(s)"))
        assert last < first, "the loss did not go down"
        if use_dist:
            print("collective backend %s, world %d%s" % (dist.get_backend(), world, " (forced)" if world == 1 else ""))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Utterance sharding across the GPUs of a node.

The reference has no distributed code; LF-MMI shards naturally because sequences are
independent (chain-computation.h:33-35).  One process per GPU; the only coupling is the
scalar sum of per-sequence log-probs (chain-computation.cc:229) and ChainLoss's frame
normaliser (loss.py:104): ONE all-reduce of a 3-float buffer per step.  Each rank
back-propagates its own [B_local,T,D] gradient shard; DDP all-reduces parameter
gradients as usual.  An all-reduce of the [B_global,T,D] gradient slab itself (what the
north-star text names) moves 10.6 GB per step at C5 - DESIGN.md §6 - so it is NOT on the
default path; `allreduce_grad_slab` provides it as an option (one fused collective:
scalars + slab) for callers that want every rank to hold the whole gradient, and bench.py
reports its cost separately.
"""
import torch
import torch.distributed as dist

__all__ = ["shard_indices", "shard_batch", "allreduce_stats", "allreduce_grad_slab", "ShardedChainLoss"]


def shard_indices(lengths, world_size, rank):
    """Global minibatch -> this rank's utterance indices.  Utterances are sorted by length
    (descending) and dealt in serpentine order (0..R-1, R-1..0, 0..R-1, ...), so every
    shard stays length-sorted (what the reference API wants, loss.py:37-40) and shards
    carry near-equal frame counts and near-equal longest sequences (kernel time follows
    the longest sequence)."""
    lengths = torch.as_tensor(lengths).cpu()
    order = torch.argsort(lengths, descending=True, stable=True)
    n = order.numel()
    pos = torch.arange(n)
    row, col = pos // world_size, pos % world_size
    owner = torch.where(row % 2 == 0, col, world_size - 1 - col)
    return order[owner == rank]


def shard_batch(x, lengths, num_graphs, world_size, rank):
    """Slice (x, lengths, num_graphs) down to this rank's shard: (x_shard, lengths_shard, num_graphs_shard, idx) with
    idx = this rank's utterance indices in the global batch.  `x` may be None when the caller materialises only its own
    rows (a data loader that reads utterances `idx`; bench.py at N > 1, where the global [B,T,D] does not exist)."""
    from .graph import ChainGraphBatch
    idx = shard_indices(lengths, world_size, rank)
    lengths = torch.as_tensor(lengths)
    xs = x.index_select(0, idx.to(x.device)) if x is not None else None
    ls = lengths.index_select(0, idx.to(lengths.device))
    gs = None
    if num_graphs is not None:
        gs = ChainGraphBatch.__new__(ChainGraphBatch)
        gs.__dict__.update(num_graphs.__dict__)
        gs._device_cache = {}
        gs.reorder(idx)
        gs.batch_size = int(idx.numel())
    return xs, ls, gs, idx


def allreduce_stats(objf, n_frames, bad_count=None, group=None, force=False):
    """[objf, n_frames, n_bad] summed over all ranks with ONE collective (RCCL all_reduce
    over xGMI on GPUs, gloo on CPU).  Returns the fp32[3] buffer; no host sync."""
    dev = objf.device
    parts = [objf.detach().to(torch.float32).reshape(1),
             torch.as_tensor(n_frames, device=dev).to(torch.float32).reshape(1),
             bad_count.to(dev).sum(dtype=torch.float32).reshape(1) if bad_count is not None
             else torch.zeros(1, dtype=torch.float32, device=dev)]
    buf = torch.cat(parts)        # one small kernel, not a fill and three copies
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


def allreduce_grad_slab(local_grad, global_idx, global_batch, stats=None, group=None, out=None):
    """OPTION (not the default path): every rank ends up with the whole [B_global,T,D]
    gradient.  Rows of different utterances are disjoint, so the all-reduce(SUM) of a slab
    that is zero except for this rank's rows is an all-gather by summation; the scalars
    (`stats`, fp32[n]) ride in the same buffer -> ONE collective per step, as the north-star
    text words it.  `global_idx` = this rank's utterance indices in the global batch
    (shard_indices); `out` = optional preallocated flat fp32 buffer of n + B_global*T*D.
    Returns (stats_sum, grad_global[B_global,T,D])."""
    Bl, T, D = local_grad.shape
    n = 0 if stats is None else int(stats.numel())
    size = n + int(global_batch) * T * D
    buf = out if out is not None else torch.empty(size, dtype=torch.float32, device=local_grad.device)
    if buf.numel() != size or buf.dtype != torch.float32:
        raise ValueError("allreduce_grad_slab: `out` must be a flat fp32 buffer of %d elements" % size)
    buf.zero_()
    if n:
        buf[:n] = stats.detach().float().reshape(-1)
    slab = buf[n:].view(int(global_batch), T, D)
    slab.index_copy_(0, torch.as_tensor(global_idx, device=local_grad.device, dtype=torch.long),
                     local_grad.detach().float())
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return (buf[:n] if n else None), slab


class _GlobalLoss(torch.autograd.Function):
    """value: the all-reduced loss [/ the all-reduced frame count]; gradient: that of the LOCAL loss, scaled alike."""

    @staticmethod
    def forward(ctx, local, stats, avg):
        ctx.denom = stats[1] if avg else None
        return stats[0] / stats[1] if avg else stats[0].clone()

    @staticmethod
    def backward(ctx, g):
        return (g if ctx.denom is None else g / ctx.denom), None, None


class ShardedChainLoss(torch.nn.Module):
    """ChainLoss over a utterance-sharded global minibatch: each rank evaluates its shard,
    the loss value is the global one (sum over ranks / global frame count when `avg`), and
    autograd yields the correctly scaled gradient for the local shard.

    `last_stats` (device fp32, never synced here) holds [global objf, global frames, global bad count] of the last step.
    With the native ChainLoss the three scalars come out of the loss call's last kernel (`loss.totals` of the tensor the call
    returned - not a class attribute: two criteria in one process do not see each other's) and are all-reduced as they are -
    no scalar kernels of the host framework around the collective; a world of one does nothing at all here unless
    `force_collective` asks for the collective anyway (one rank under torch.distributed: the RCCL leg of the GPU tests)."""

    def __init__(self, den_graph, leaky_coefficient=1e-5, avg=True, group=None, loss_cls=None, force_collective=False):
        super().__init__()
        self._native = loss_cls is None or bool(getattr(loss_cls, "reports_bad_count", False))
        if loss_cls is None:
            from .loss import ChainLoss as loss_cls
        self.local = loss_cls(den_graph, leaky_coefficient, avg=False)
        self.avg = avg
        self.group = group
        self.force_collective = bool(force_collective)
        self.last_stats = None

    def _world(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def forward(self, x, x_lengths, num_graphs):
        local = self.local(x, x_lengths, num_graphs)                  # sum over local utterances
        # [loss, frames, bad, ...] of THIS call, on the device: they came back with the tensor (pychain_amd/loss.py: _attach)
        totals = getattr(local, "totals", None) if self._native else None
        collective = dist.is_available() and dist.is_initialized() and (self._world() > 1 or self.force_collective)
        if totals is not None:
            if not collective:
                self.last_stats = totals[:3]
                return local / totals[1] if self.avg else local
            stats = totals[:3].clone()
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.group)
            self.last_stats = stats
            return _GlobalLoss.apply(local, stats, self.avg)
        frames = torch.as_tensor(x_lengths).sum()
        bad = getattr(local, "bad_count", None) if self._native else None   # the reference's `ok` of every rank rides along
        if isinstance(bad, (tuple, list)):                            # (a two-call loss: denominator's and numerator's)
            bad = torch.cat([b.reshape(-1) for b in bad if b is not None]) if any(b is not None for b in bad) else None
        stats = self.last_stats = allreduce_stats(local, frames, bad, self.group, force=self.force_collective)
        # value: global; gradient: d(local)/dx scaled by the global normaliser
        denom = stats[1] if self.avg else torch.ones((), device=stats.device)
        return (local - local.detach() + stats[0]) / denom

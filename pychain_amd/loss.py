"""ChainFunction / ChainLoss: the LF-MMI autograd API of pychain on MI355X.

Mirror of the reference's pychain/loss.py (same names, signatures, defaults and
error behaviour).  What differs is underneath:

  * no extra passes over [B,T,D]: clamp(-30,30) and exp (loss.py:30,43) are fused
    into the HIP kernels, the log-domain gradient is produced in the linear
    domain directly (loss.py:77);
  * the denominator graph is compiled once to a device-resident plan that all B
    sequences share (no B-fold replication, no per-call H2D; graph.py:99-120,
    chain-computation.cc:77-89);
  * lengths may be given in any order and on any device (the reference needs
    them sorted descending on the CPU for pack_padded_sequence, loss.py:37-40);
  * fp16 / bf16 network outputs are accepted and read by the kernels AS THEY ARE (converted where they land, all
    arithmetic in fp32; the gradient is rounded to the input's dtype where it is written: no fp32 copy of [B,T,D]);
    the reference's C++ accessors take float32 only.

CPU tensors are served by the library's host twins (csrc/cpu.cpp), device tensors by the HIP kernels - never one for the
other: a device tensor that cannot reach the kernels raises.
"""
import torch
import torch.nn as nn

from . import _lib, _plan, native
from .graph import ChainGraphBatch

__all__ = ["ChainFunction", "ChainLossFunction", "ChainLoss"]


class ChainFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, input_lengths, graphs, leaky_coefficient=1e-5):
        B = input.size(0)
        if B != graphs.batch_size:
            raise ValueError(
                "input batch size ({}) does not equal to graph batch size ({})"
                .format(B, graphs.batch_size))
        x = input.detach()
        report = {}
        objf, input_grad, bad = ChainFunction._occupancies(x, input_lengths, graphs, leaky_coefficient, totals=True, report=report)
        return ChainFunction._forward_tail(ctx, input, x, input_lengths, graphs, leaky_coefficient, objf, input_grad, bad,
                                           report.get("totals"))

    @staticmethod
    def _occupancies(x, input_lengths, graphs, leaky_coefficient, totals=False, report=None):
        """(objf per sequence, occupancies = gradient for an upstream gradient of 1, bad_count).  `totals`: the
        denominator's last kernel also leaves [sum objf, frames, bad, sum objf, ...] (include/pychain_hip.h: totals) in
        `report["totals"]` - the caller attaches them to the tensor it returns - and the sum is returned in place of the
        per-sequence values (no reduction launch behind the call)."""
        D = x.size(2)
        if not x.is_cuda:
            # CPU tensors: the library's host twins (pychain_amd/csrc/cpu.cpp) - what the reference does with them
            # (chain-computation.cc:40,136-175); device tensors never come here
            return native.cpu_forward_backward(graphs, x, input_lengths, leaky_coefficient)
        if not graphs.log_domain:   # usually the denominator
            if graphs.shared_graph is not None:
                plan = _plan.graph_plan(graphs.shared_graph, D, x.device)
            else:
                names = ("forward_transitions", "forward_transition_indices", "forward_transition_probs",
                         "backward_transitions", "backward_transition_indices", "backward_transition_probs",
                         "leaky_probs", "initial_probs", "final_probs")
                # keyed like graph_plan / device_tensors: an in-place edit or a replaced tensor re-compiles
                key = ("den_plans", str(x.device), D) + tuple(
                    (getattr(graphs, n).data_ptr(), getattr(graphs, n)._version) for n in names)
                hit = graphs._device_cache.get(key)
                if hit is None:
                    for k in [k for k in graphs._device_cache if k[:1] == ("den_plans",)]:
                        del graphs._device_cache[k]
                    hit = _plan.batch_plans({n: getattr(graphs, n) for n in names}, D, x.device)
                    graphs._device_cache[key] = hit
                plan = hit
            if totals:
                objf, input_grad, bad, tot = native.den_forward_backward(
                    plan, x, input_lengths, leaky_coefficient, input_is_exp=False, totals=True)
                if report is not None:
                    report["totals"] = tot
                objf = native.totals_scalar(tot)   # (no launch; not a view of the statistics)
            else:
                objf, input_grad, bad = native.den_forward_backward(
                    plan, x, input_lengths, leaky_coefficient, input_is_exp=False)
        else:                       # usually the numerator
            gt = graphs.device_tensors(x.device)
            gstride = 0 if graphs.shared_graph is not None else 1
            objf, input_grad, bad = native.num_forward_backward(
                gt, gstride, graphs.num_states, x, input_lengths, grad_mode=_lib.GRAD_LINEAR)
        return objf, input_grad, bad

    @staticmethod
    def _forward_tail(ctx, input, x, input_lengths, graphs, leaky_coefficient, objf, input_grad, bad, tot=None):
        # The occupancies are the gradient for an upstream gradient of 1.  backward() scales the
        # buffer in place on the device (a no-op launch when the upstream gradient is exactly 1,
        # i.e. `objf.backward()`) and hands it to autograd, instead of the reference's extra
        # read+write pass over [B,T,D] (torch.mul, loss.py:85).  A SECOND backward over the same graph
        # (retain_graph=True; the reference allows it, loss.py:82-87) finds the buffer gone - autograd
        # owns it, it may be x.grad by now - and evaluates the occupancies again from the inputs kept
        # here by reference.  `retain_grad_buffer` = True keeps a private copy instead (reference cost).
        if ChainFunction.retain_grad_buffer:
            ctx.save_for_backward(input_grad)
        else:
            ctx.grad_buf = input_grad
            ctx.again = _recompute(x, lambda: ChainFunction._occupancies(x, input_lengths, graphs, leaky_coefficient),
                                   lambda r: (r[1], r[2]))
        ctx.in_dtype = input.dtype   # fp16 / bf16 inputs are evaluated in fp32; the gradient goes back in their dtype
        ctx.bad_count = bad          # device int32[1]; the reference's `ok`, never synced here
        out = objf.sum() if objf.dim() else objf               # (0-dim: the sum came with the call)
        return _attach(out, tot, bad)

    retain_grad_buffer = False
    # DEPRECATED mirrors of what the LAST call in this process reported (any criterion, any thread: two losses in one process
    # overwrite each other's).  Read `loss.totals` / `loss.totals_all` / `loss.bad_count` of the tensor a call RETURNED instead.
    last_totals = None           # device float[4] (include/pychain_hip.h: totals), None where the call produced none
    last_totals_all = None       # ... all eight: [5..7] say how a time-segmented call went (segments redone, count, worst mismatch)
    last_bad_count = None

    @staticmethod
    def backward(ctx, objf_grad):
        # clamp is inside the Function and therefore not differentiated (loss.py:30,82-87)
        if ctx.saved_tensors:
            input_grad, = ctx.saved_tensors
            return torch.mul(input_grad, objf_grad).to(ctx.in_dtype), None, None, None
        grad = _take_grad_buffer(ctx, "grad_buf")
        if grad is None:
            grad, ctx.bad_count = ctx.again()                  # second backward over a retained graph: evaluate again
            ChainFunction.last_bad_count = ctx.bad_count
        if not grad.is_cuda:
            return torch.mul(grad, objf_grad).to(ctx.in_dtype), None, None, None          # (loss.py:85, as it is)
        return native.rescale_(grad, objf_grad).to(ctx.in_dtype), None, None, None


def _attach(out, totals, bad):
    """What a call reports travels WITH the tensor it returns - `out.totals` (device float[4]: loss, frames, bad count, sum den - sum
    num; None where the call produced none), `out.totals_all` (all eight of include/pychain_hip.h), `out.bad_count` (device
    int32, the reference's `ok` as a count; never synced here) - so that two criteria, or two host threads, in one process do
    not read each other's.  The class attributes ChainFunction.last_* remain as deprecated mirrors."""
    out.totals = None if totals is None else totals[:4]
    out.totals_all = totals
    out.bad_count = bad
    ChainFunction.last_totals, ChainFunction.last_totals_all, ChainFunction.last_bad_count = out.totals, totals, bad
    return out


def _recompute(x, evaluate, pick):
    """The closure a second backward over a retained graph calls: evaluates again from the inputs kept by reference.
    autograd's version check only guards tensors saved with save_for_backward, so it is restated here: an in-place
    edit of the network output between the two backward calls would silently change the gradient (the reference
    saved the gradient itself, loss.py:79)."""
    version = x._version

    def again():
        if x._version != version:
            raise RuntimeError(
                "one of the variables needed for gradient computation has been modified by an inplace operation: "
                "the network output given to the LF-MMI loss is at version %d; expected version %d (second backward "
                "over a retained graph re-evaluates the loss from it)" % (x._version, version))
        return pick(evaluate())
    return again


def _take_grad_buffer(ctx, attr):
    """The gradient buffer written in forward, handed over ONCE (autograd then owns it: a leaf's
    .grad takes it without a copy); None when it is gone."""
    buf = getattr(ctx, attr, None)
    setattr(ctx, attr, None)
    return buf


class ChainLossFunction(torch.autograd.Function):
    """Denominator + numerator in one pass, split at the autograd
    boundary: forward runs the four recursions (numerator on a side stream) and returns the
    loss; backward runs the time-parallel occupancy passes and writes
    (gamma_den - gamma_num) * objf_grad [/ frames] ONCE, with the upstream gradient read on
    the device - instead of two dense gradients, two scalar multiplies and an autograd add
    (loss.py:85,100-104).  Same numbers as the two-call path."""

    @staticmethod
    def forward(ctx, input, input_lengths, den_graph, num_graphs, leaky_coefficient, avg):
        x = input.detach()
        B, D = x.size(0), x.size(2)
        if B != num_graphs.batch_size:
            raise ValueError(
                "input batch size ({}) does not equal to graph batch size ({})"
                .format(B, num_graphs.batch_size))
        lengths = torch.as_tensor(input_lengths)
        plan = _plan.graph_plan(den_graph, D, x.device)
        gt = num_graphs.device_tensors(x.device)
        gstride = 0 if num_graphs.shared_graph is not None else 1
        # avg=True divides by the frame count (loss.py:103-104): a host scalar when the lengths
        # live on the host, else a device scalar - never a sync
        ctx.host_scale, ctx.dev_norm = 1.0, None
        if avg:
            if lengths.is_cuda:
                ctx.dev_norm = lengths.sum().to(torch.float32)
            else:
                ctx.host_scale = 1.0 / float(lengths.sum())
        # When a gradient will be asked for, the occupancy passes run inside forward, overlapped
        # with the recursions, for an upstream gradient of 1 (what `loss.backward()` sends);
        # backward then only rescales if the upstream gradient turns out to differ.
        ctx.speculative = bool(ctx.needs_input_grad[0]) and ChainLossFunction.overlap
        # (2-byte network outputs go to the kernels as they are when the gradient is written here, or never)
        half_ok = ctx.speculative or not bool(ctx.needs_input_grad[0])
        den_objf, num_objf, bad, state, totals = native.chain_loss_forward(
            plan, gt, gstride, num_graphs.num_states, x, lengths, leaky_coefficient,
            with_grad=ctx.speculative, grad_scale=ctx.host_scale, loss_scale=ctx.host_scale, norm_dev=ctx.dev_norm,
            half_ok=half_ok)
        # -(num - den) [/ frames], loss.py:100-104, comes with the call (the last workgroup of its last kernel adds the
        # per-sequence objectives up): no reduction / subtraction / scaling launches behind it
        objf = native.totals_scalar(totals)    # (no launch; not a view of the statistics: `loss /= n` works)
        ctx.state = state
        # a second backward over a retained graph (loss.py:82-87 allows it) runs the recursions again
        spec, hscale, dnorm = ctx.speculative, ctx.host_scale, ctx.dev_norm      # (locals: the closure must not hold ctx)
        ctx.again = _recompute(x, lambda: native.chain_loss_forward(
            plan, gt, gstride, num_graphs.num_states, x, lengths, leaky_coefficient,
            with_grad=spec, grad_scale=hscale, norm_dev=dnorm, half_ok=half_ok), lambda r: (r[3], r[2]))      # (state, bad)
        ctx.in_dtype = input.dtype
        ctx.bad_count = bad                      # int32[2]: denominator, numerator; never synced here
        return _attach(objf, totals, bad)

    overlap = True     # class-level switch: False = occupancy passes run in backward (no speculation)

    @staticmethod
    def backward(ctx, objf_grad):
        state = _take_grad_buffer(ctx, "state")
        if state is None:
            state, ctx.bad_count = ctx.again()
            ChainFunction.last_bad_count = ctx.bad_count
        if ctx.speculative:
            # (a device-side normaliser - avg=True with the lengths on the device - was divided into the gradient by the call
            # that wrote it: include/pychain_hip.h, loss_norm_dev; an upstream gradient of exactly 1 then costs one tiny launch)
            grad = native.rescale_(state.grad, objf_grad)
        else:
            g = objf_grad if ctx.dev_norm is None else objf_grad / ctx.dev_norm.to(objf_grad.device)
            grad, bad = native.chain_loss_backward(state, ctx.host_scale, g)
            ctx.bad_count = ctx.bad_count + bad              # (the occupancy launches' own checks)
            ChainFunction.last_bad_count = ctx.bad_count
        state.grad = None         # the stored trajectories go with `state`
        state.den_ws = state.num_ws = None
        return grad.to(ctx.in_dtype), None, None, None, None, None


class ChainLoss(nn.Module):
    def __init__(self, den_graph, leaky_coefficient=1e-5, avg=True):
        super(ChainLoss, self).__init__()
        self.den_graph = den_graph
        self.avg = avg
        self.leaky_coefficient = leaky_coefficient
        self.fused = True   # one-pass kernel path; False = two ChainFunction calls as in the reference

    def forward(self, x, x_lengths, num_graphs):
        if (self.fused and x.is_cuda and not self.den_graph.log_domain and num_graphs.log_domain):
            return ChainLossFunction.apply(x, x_lengths, self.den_graph, num_graphs,
                                           self.leaky_coefficient, self.avg)
        batch_size = x.size(0)
        den_graphs = ChainGraphBatch(self.den_graph, batch_size)
        den_objf = ChainFunction.apply(x, x_lengths, den_graphs, self.leaky_coefficient)
        num_objf = ChainFunction.apply(x, x_lengths, num_graphs)
        objf = -(num_objf - den_objf)
        if self.avg:
            objf = objf / x_lengths.sum()
        # (two native calls made this loss: neither's totals are the step's - ShardedChainLoss finds none and all-reduces its own
        # three scalars; the two bad counts ride along as they are: no launch here)
        objf.totals = objf.totals_all = None
        objf.bad_count = (den_objf.bad_count, num_objf.bad_count)
        ChainFunction.last_totals = ChainFunction.last_totals_all = None
        return objf

"""Tensor-level entry points over the C ABI: the drop-in for the reference's
`pychain_C` module (pytorch_binding/src/pychain.cc:131-135).

`forward_backward` / `forward_backward_log_domain` / `set_verbose_level` keep the
reference's positional signatures and return `[objf, grad, ok]`; the `den_*` /
`num_*` functions are what `pychain_amd.loss` calls (device-resident plans, fused
clamp/exp, per-sequence objf).
"""
import torch

from . import _lib, _plan

import threading

_ws_cache = {}          # (device, stream, tag) -> uint8 tensor, most recently used last
_ws_lock = threading.Lock()      # (criteria on several host threads share this module)
_WS_CACHE_ENTRIES = 8   # a training process uses 2 (den, num) per stream; more are streams that came and went


def _require_device(t, what):
    if not t.is_cuda:
        raise RuntimeError(
            "pychain_amd: %s must live on a HIP device (got %s): the HIP entry points have no CPU fallback "
            "(CPU tensors are served by cpu_forward_backward, never the other way round)." % (what, t.device))


# ---------------------------------------------------------------------------
# CPU tensors: the library's host twins (include/pychain_hip.h: pychain_hip_cpu_*; pychain_amd/csrc/cpu.cpp).  The reference
# computes on whatever device its input lives on (chain-computation.cc:40): so does this package - CPU tensors HERE, device
# tensors in the HIP kernels, and never one for the other (a device tensor that cannot reach the kernels raises).
# ---------------------------------------------------------------------------
CPU_THREADS = 0          # host threads the sequences of a minibatch are dealt to (0 = one per hardware thread)
_GRAPH_INT = ("forward_transitions", "forward_transition_indices", "backward_transitions", "backward_transition_indices")


def _cpu_graph(graphs, with_leaky):
    """(tensors in ABI order, graph_batch_stride): one graph for every sequence (stride 0) or [B,...] tensors."""
    src, stride = (graphs.shared_graph, 0) if graphs.shared_graph is not None else (graphs, 1)
    names = ["forward_transitions", "forward_transition_indices", "forward_transition_probs",
             "backward_transitions", "backward_transition_indices", "backward_transition_probs"]
    names += (["leaky_probs"] if with_leaky else []) + ["initial_probs", "final_probs"]
    ts = [getattr(src, n).to(torch.int32 if n in _GRAPH_INT else torch.float32).contiguous() for n in names]
    return ts, stride


def cpu_forward_backward(graphs, x, lengths, leaky_coefficient=1e-5, input_is_exp=False, grad_mode=_lib.GRAD_LINEAR, clamp=True):
    """ChainFunction's computation on CPU tensors: (objf_per_seq[B], grad[B,T,D] fp32, bad_count int32[1]).  Denominator
    (probability-domain graphs) or numerator (log-domain graphs) by `graphs.log_domain`.  `clamp` False (numerator): the network
    output as it is - what pychain_C.forward_backward_log_domain computes on (the reference clamps in Python, loss.py:30)."""
    if x.is_cuda:
        raise RuntimeError("pychain_amd: cpu_forward_backward is for CPU tensors; device tensors run on the HIP kernels")
    xf = x.detach().to(torch.float32).contiguous()
    B, T, D = xf.shape
    lc = torch.as_tensor(lengths).to(torch.int64).cpu().contiguous()
    _check_lengths(lc, B, T)
    ts, stride = _cpu_graph(graphs, not graphs.log_domain)
    H, K = int(ts[1].shape[-2]), int(ts[0].shape[-2])
    objf = torch.empty(B, dtype=torch.float32)
    grad = torch.empty(B, T, D, dtype=torch.float32)
    bad = torch.zeros(1, dtype=torch.int32)
    L = _lib.lib()
    ptrs = [t.data_ptr() for t in ts]
    if not graphs.log_domain:
        _lib.check(L.pychain_hip_cpu_den_forward_backward(
            *ptrs, stride, xf.data_ptr(), int(bool(input_is_exp)), lc.data_ptr(), B, T, D, H, K,
            float(leaky_coefficient), 1.0, objf.data_ptr(), grad.data_ptr(), bad.data_ptr(), int(CPU_THREADS)),
            "pychain_hip_cpu_den_forward_backward")
    else:
        _lib.check(L.pychain_hip_cpu_num_forward_backward(
            *ptrs, stride, xf.data_ptr(), lc.data_ptr(), B, T, D, H, K, int(grad_mode) | (0 if clamp else _lib.CPU_NO_CLAMP), 1.0,
            objf.data_ptr(), grad.data_ptr(), bad.data_ptr(), int(CPU_THREADS)), "pychain_hip_cpu_num_forward_backward")
    return objf, grad, bad


def _workspace(nbytes, device, tag="a"):
    """Scratch for the stored trajectories, reused from call to call.  Keyed by the CALLER'S STREAM as well:
    calls on one stream are ordered, so the next one may overwrite what the previous one left; calls on
    two streams of one device run concurrently and must not share it.  (A replaced buffer goes back to the
    caching allocator, which hands it out again only in the stream order of its allocation.)"""
    key = (str(device), _stream(device), tag)
    with _ws_lock:
        ws = _ws_cache.pop(key, None)
        if ws is None or ws.numel() < nbytes:
            ws = None                      # (the old buffer goes back to the allocator before the larger one is asked for)
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws_cache[key] = ws                # (re-)inserted last = most recently used
        while len(_ws_cache) > _WS_CACHE_ENTRIES:
            _ws_cache.pop(next(iter(_ws_cache)))       # least recently used: multi-GB buffers of streams no longer in use
    return ws


def release_workspaces():
    """Drop every cached workspace (several GB at C3 sizes) and the memoised plans / graph uploads of the
    pychain_C-compatible surface."""
    with _ws_lock:
        _ws_cache.clear()
        _compat_cache.clear()


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def _lengths_dev(lengths, device):
    return torch.as_tensor(lengths).to(device=device, dtype=torch.int64, non_blocking=True).contiguous()


def _check_lengths(lengths, B, T):
    lc = torch.as_tensor(lengths)
    if lc.numel() != B:
        raise ValueError("sequence_lengths has %d entries for a batch of %d" % (lc.numel(), B))
    if not lc.is_cuda:   # cheap host-side validation only when it costs no sync
        if int(lc.min()) < 1 or int(lc.max()) > T:
            raise ValueError("sequence lengths must be in [1, %d]" % T)


# (Time segments - include/pychain_hip.h: totals[5..7], DESIGN.md §3.13: the burn-in controller that used to live here, fed by
# non-blocking copies of totals[5] whenever their event happened to have fired, is in the LIBRARY since ABI 16 - a device-
# resident state per plan, read and updated by the call's own kernels in stream order: pychain_amd/_plan.py attaches it, nothing
# on the host reads it, and the same run gives the same bits.)


HALF_ROWS = True        # False: bf16 / fp16 network outputs are up-cast on the host side of the ABI (the tests compare the two ways)
_DTYPE_CODE = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}


def _rows_as_given(x, native):
    """(tensor the kernels read, its PYCHAIN_HIP_* element code): a bf16 / fp16 network output goes to the kernels as it is
    where the call's kernel family takes 2-byte rows (`native`: the library's own answer, include/pychain_hip.h:
    pychain_hip_*_half_native) - no fp32 copy of [B,T,D], and the gradient comes back in the same type; otherwise it is
    up-cast here, once (general plans, the two-barrier recursion, rows that are not a multiple of 8 pdfs, ...)."""
    if x.dtype == torch.float32:
        return x, _lib.F32
    if HALF_ROWS and x.dtype in _DTYPE_CODE and native():
        return x, _DTYPE_CODE[x.dtype]
    return x.float(), _lib.F32


def den_forward_backward(plan, x, lengths, leaky_coefficient=1e-5, input_is_exp=False, grad_scale=1.0, totals=False):
    """Denominator on the GPU.  `plan`: _plan.DevicePlan.
    Returns (objf_per_seq[B], grad[B,T,D], bad_count[1]) and, `totals`, the device float[4] of
    include/pychain_hip.h (sum of the objectives, frames, bad count - from the call's last kernel, no extra launch)."""
    _require_device(x, "nnet_output")
    num_states = plan.num_states
    x = x.contiguous()
    B, T, D = x.shape
    _check_lengths(lengths, B, T)
    L = _lib.lib()
    dev = x.device
    with torch.cuda.device(dev):
        x, xcode = _rows_as_given(x, lambda: L.pychain_hip_den_half_native(plan.stride, plan.slot_rows, int(num_states), D, B, T))
        ld = _lengths_dev(lengths, dev)
        objf = torch.empty(B, dtype=torch.float32, device=dev)
        grad = torch.empty_like(x)                 # (in the type the kernels read: the gradient is rounded where it is written)
        bad = torch.empty(1, dtype=torch.int32, device=dev)
        tot = torch.empty(_lib.TOTALS, dtype=torch.float32, device=dev)
        # (the [B,T,D] buffer of the rows exp'd ahead only where this call will use it: ADVICE r4)
        full = L.pychain_hip_den_uses_row_buffer(plan.stride, plan.slot_rows, int(num_states), D, B, T, int(bool(input_is_exp)))
        nws = (L.pychain_hip_den_workspace_bytes if full else L.pychain_hip_den_workspace_min_bytes)(B, T, int(num_states), D)
        ws = _workspace(nws, dev, "den")
        _lib.check(L.pychain_hip_den_forward_backward(
            plan.blob.data_ptr(), plan.stride, plan.slot_rows, int(num_states), D, x.data_ptr(), xcode,
            int(bool(input_is_exp)),
            ld.data_ptr(), B, T, float(leaky_coefficient), float(grad_scale),
            objf.data_ptr(), grad.data_ptr(), bad.data_ptr(), tot.data_ptr(),
            ws.data_ptr(), ws.numel(), _stream(dev)),
            "pychain_hip_den_forward_backward")
    return (objf, grad, bad, tot) if totals else (objf, grad, bad)


def num_forward_backward(gt, graph_stride, num_states, x, lengths, grad_mode=_lib.GRAD_LINEAR,
                         grad_scale=1.0, grad_out=None):
    """Numerator on the GPU.  `gt`: dict of device graph tensors.  Returns
    (objf_per_seq[B], grad[B,T,D], bad_count[1])."""
    _require_device(x, "nnet_output")
    x = x.contiguous()
    B, T, D = x.shape
    _check_lengths(lengths, B, T)
    K = gt["forward_transitions"].shape[1]
    L = _lib.lib()
    dev = x.device
    with torch.cuda.device(dev):
        x, xcode = _rows_as_given(x, lambda: L.pychain_hip_num_half_native(int(num_states), K, D))
        ld = _lengths_dev(lengths, dev)
        objf = torch.empty(B, dtype=torch.float32, device=dev)
        if grad_mode == _lib.GRAD_ACCUM:
            if grad_out is None:
                raise ValueError("GRAD_ACCUM needs grad_out")
            grad = grad_out
        else:
            grad = torch.empty(x.shape, dtype=torch.float32, device=dev)     # (this entry point's gradient is fp32 whatever it read)
        bad = torch.empty(1, dtype=torch.int32, device=dev)
        nws = L.pychain_hip_num_workspace_bytes(B, T, int(num_states), K, D)
        ws = _workspace(nws, dev, "num")
        _lib.check(L.pychain_hip_num_forward_backward(
            gt["forward_transitions"].data_ptr(), gt["forward_transition_indices"].data_ptr(),
            gt["forward_transition_probs"].data_ptr(), gt["backward_transitions"].data_ptr(),
            gt["backward_transition_indices"].data_ptr(), gt["backward_transition_probs"].data_ptr(),
            gt["initial_probs"].data_ptr(), gt["final_probs"].data_ptr(), int(graph_stride),
            x.data_ptr(), xcode, ld.data_ptr(), B, T, D, int(num_states), K, int(grad_mode), float(grad_scale),
            objf.data_ptr(), grad.data_ptr(), bad.data_ptr(), ws.data_ptr(), ws.numel(), _stream(dev)),
            "pychain_hip_num_forward_backward")
    return objf, grad, bad


def chain_loss_forward_backward(plan, gt, graph_stride, num_states_num, x, lengths,
                                leaky_coefficient=1e-5, grad_scale=1.0):
    """Fused ChainLoss: returns (den_objf[B], num_objf[B], grad[B,T,D] = grad_scale*(gamma_den - gamma_num),
    bad_count[2]).  The numerator recursion overlaps the denominator on a side stream."""
    _require_device(x, "nnet_output")
    x = x.contiguous()
    B, T, D = x.shape
    _check_lengths(lengths, B, T)
    K = gt["forward_transitions"].shape[1]
    L = _lib.lib()
    dev = x.device
    with torch.cuda.device(dev):
        x, xcode = _rows_as_given(x, lambda: L.pychain_hip_chain_loss_half_native(
            plan.stride, plan.slot_rows, plan.num_states, D, B, T, int(num_states_num), K))
        ld = _lengths_dev(lengths, dev)
        den_objf = torch.empty(B, dtype=torch.float32, device=dev)
        num_objf = torch.empty(B, dtype=torch.float32, device=dev)
        grad = torch.empty_like(x)
        bad = torch.empty(2, dtype=torch.int32, device=dev)
        dws = _workspace(L.pychain_hip_den_workspace_min_bytes(B, T, plan.num_states, D), dev, "den")   # (the fused loss never exp's rows ahead)
        nws = _workspace(L.pychain_hip_num_workspace_bytes(B, T, int(num_states_num), K, D), dev, "num")
        _lib.check(L.pychain_hip_chain_loss_forward_backward(
            plan.blob.data_ptr(), plan.stride, plan.slot_rows, plan.num_states, float(leaky_coefficient),
            gt["forward_transitions"].data_ptr(), gt["forward_transition_indices"].data_ptr(),
            gt["forward_transition_probs"].data_ptr(), gt["backward_transitions"].data_ptr(),
            gt["backward_transition_indices"].data_ptr(), gt["backward_transition_probs"].data_ptr(),
            gt["initial_probs"].data_ptr(), gt["final_probs"].data_ptr(), int(graph_stride),
            int(num_states_num), K,
            x.data_ptr(), xcode, ld.data_ptr(), B, T, D, float(grad_scale),
            den_objf.data_ptr(), num_objf.data_ptr(), grad.data_ptr(), bad.data_ptr(), 1.0, 0, 0,
            dws.data_ptr(), dws.numel(), nws.data_ptr(), nws.numel(), _stream(dev)),
            "pychain_hip_chain_loss_forward_backward")
    return den_objf, num_objf, grad, bad


class ChainLossState(object):
    """What `chain_loss_forward` leaves behind for `chain_loss_backward`: the stored
    trajectories (workspaces) and the handles of everything the occupancy passes read."""
    __slots__ = ("plan", "gt", "graph_stride", "num_states_num", "x", "lengths_dev", "den_ws", "num_ws", "shape",
                 "grad", "num_compat")


def chain_loss_forward(plan, gt, graph_stride, num_states_num, x, lengths, leaky_coefficient=1e-5,
                       with_grad=False, grad_scale=1.0, loss_scale=1.0, norm_dev=None, half_ok=True):
    """Recursions (and, `with_grad`, the occupancy passes overlapped with them: state.grad =
    grad_scale * (gamma_den - gamma_num)).  Returns (den_objf[B], num_objf[B], bad_count[2], state, totals) with
    totals = device float[4] [(sum den - sum num) * loss_scale [/ norm_dev], frames, bad count, sum den - sum num]
    written by the call's last kernel (include/pychain_hip.h)."""
    _require_device(x, "nnet_output")
    x = x.contiguous()
    B, T, D = x.shape
    _check_lengths(lengths, B, T)
    K = gt["forward_transitions"].shape[1]
    L = _lib.lib()
    dev = x.device
    st = ChainLossState()
    with torch.cuda.device(dev):
        # (`half_ok` False: a later chain_loss_backward on this state - it accumulates the numerator into an fp32 gradient)
        x, xcode = _rows_as_given(x, lambda: half_ok and L.pychain_hip_chain_loss_half_native(
            plan.stride, plan.slot_rows, plan.num_states, D, B, T, int(num_states_num), K))
        ld = _lengths_dev(lengths, dev)
        den_objf = torch.empty(B, dtype=torch.float32, device=dev)
        num_objf = torch.empty(B, dtype=torch.float32, device=dev)
        bad = torch.empty(2, dtype=torch.int32, device=dev)
        totals = torch.empty(_lib.TOTALS, dtype=torch.float32, device=dev)
        if norm_dev is not None:
            norm_dev = norm_dev.detach().to(device=dev, dtype=torch.float32).contiguous()
        # per-call workspaces (they must survive until backward); the caching allocator makes this cheap
        dws = torch.empty(L.pychain_hip_den_workspace_min_bytes(B, T, plan.num_states, D), dtype=torch.uint8, device=dev)
        nws = torch.empty(L.pychain_hip_num_workspace_bytes(B, T, int(num_states_num), K, D), dtype=torch.uint8,
                          device=dev)
        grad = torch.empty_like(x) if with_grad else None
        _lib.check(L.pychain_hip_chain_loss_forward(
            plan.blob.data_ptr(), plan.stride, plan.slot_rows, plan.num_states, float(leaky_coefficient),
            gt["forward_transitions"].data_ptr(), gt["forward_transition_indices"].data_ptr(),
            gt["forward_transition_probs"].data_ptr(), gt["backward_transitions"].data_ptr(),
            gt["backward_transition_indices"].data_ptr(), gt["backward_transition_probs"].data_ptr(),
            gt["initial_probs"].data_ptr(), gt["final_probs"].data_ptr(), int(graph_stride),
            int(num_states_num), K, x.data_ptr(), xcode, ld.data_ptr(), B, T, D,
            den_objf.data_ptr(), num_objf.data_ptr(), grad.data_ptr() if with_grad else 0, float(grad_scale),
            bad.data_ptr(), float(loss_scale), 0 if norm_dev is None else norm_dev.data_ptr(), totals.data_ptr(),
            dws.data_ptr(), dws.numel(), nws.data_ptr(), nws.numel(), _stream(dev)),
            "pychain_hip_chain_loss_forward")
    st.grad = grad
    # (which numerator wrote the stored rows: backward runs on autograd's thread, where the caller's thread options do not reach)
    st.num_compat = _lib.get_option("num_compat") or "0"
    st.plan, st.gt, st.graph_stride, st.num_states_num = plan, gt, int(graph_stride), int(num_states_num)
    st.x, st.lengths_dev, st.den_ws, st.num_ws, st.shape = x, ld, dws, nws, (B, T, D, K)
    return den_objf, num_objf, bad, st, totals


def chain_loss_backward(st, grad_scale=1.0, grad_scale_dev=None):
    """Occupancy passes: grad[B,T,D] = grad_scale * grad_scale_dev * (gamma_den - gamma_num), written once.
    `grad_scale_dev`: optional 0-dim float32 device tensor (the upstream autograd gradient)."""
    B, T, D, K = st.shape
    L = _lib.lib()
    dev = st.x.device
    with torch.cuda.device(dev):
        grad = torch.empty_like(st.x)
        bad = torch.empty(2, dtype=torch.int32, device=dev)
        sptr = 0
        if grad_scale_dev is not None:
            grad_scale_dev = grad_scale_dev.detach().to(device=dev, dtype=torch.float32).contiguous()
            sptr = grad_scale_dev.data_ptr()
        with _lib.option("num_compat", st.num_compat):
            _lib.check(L.pychain_hip_chain_loss_backward(
                st.plan.blob.data_ptr(), st.plan.stride, st.plan.slot_rows, st.plan.num_states,
                st.gt["forward_transitions"].data_ptr(), st.gt["forward_transition_indices"].data_ptr(),
                st.gt["forward_transition_probs"].data_ptr(),
                st.graph_stride, st.num_states_num, K, st.x.data_ptr(), _DTYPE_CODE[st.x.dtype], st.lengths_dev.data_ptr(), B, T, D,
                float(grad_scale), sptr, grad.data_ptr(), bad.data_ptr(),
                st.den_ws.data_ptr(), st.den_ws.numel(), st.num_ws.data_ptr(), st.num_ws.numel(), _stream(dev)),
                "pychain_hip_chain_loss_backward")
    return grad, bad


def loss_total(den_objf, num_objf, scale=1.0, norm_dev=None):
    """(den_objf.sum() - num_objf.sum()) * scale [/ norm_dev] as a 0-dim device tensor, one launch
    (the scalar of ChainLoss.forward, loss.py:100-104)."""
    dev = den_objf.device
    out = torch.empty((), dtype=torch.float32, device=dev)
    if norm_dev is not None:
        norm_dev = norm_dev.detach().to(device=dev, dtype=torch.float32).contiguous()
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().pychain_hip_loss_total(
            den_objf.data_ptr(), 0 if num_objf is None else num_objf.data_ptr(), den_objf.numel(), float(scale),
            0 if norm_dev is None else norm_dev.data_ptr(), out.data_ptr(), _stream(dev)), "pychain_hip_loss_total")
    return out


def totals_scalar(totals):
    """The call's scalar (totals[4], the second copy of totals[0]: include/pychain_hip.h) as a 0-dim tensor that is NOT an
    autograd view and shares no element with the statistics totals[:4]: a caller may `loss /= n` in place, as it may with
    the fresh tensor the reference returns (a view created inside a custom Function refuses in-place ops)."""
    return torch.empty((), dtype=totals.dtype, device=totals.device).set_(totals.untyped_storage(), totals.storage_offset() + 4, ())


def rescale_(t, scale_dev):
    """t *= scale_dev (0-dim device tensor), skipped on the device when the scalar is exactly 1."""
    scale_dev = scale_dev.detach().to(device=t.device, dtype=torch.float32).contiguous()
    with torch.cuda.device(t.device):
        _lib.check(_lib.lib().pychain_hip_rescale(t.data_ptr(), _DTYPE_CODE[t.dtype], t.numel(), scale_dev.data_ptr(), _stream(t.device)),
                   "pychain_hip_rescale")
    return t


# ---------------------------------------------------------------------------
# pychain_C-compatible surface (positional signatures of pychain.cc:26-41, :81-94)
# ---------------------------------------------------------------------------
_GRAPH6 = ["forward_transitions", "forward_transition_indices", "forward_transition_probs",
           "backward_transitions", "backward_transition_indices", "backward_transition_probs"]


def _check_contiguous(**named):
    """CHECK_CONTIGUOUS of pychain.cc:24,42-54,95-107: every tensor argument, RuntimeError with the
    reference's message.  One extension: the stride-0 batch views of `ChainGraphBatch(one_graph, B)`
    (the reference materialises them with .repeat, graph.py:104-119; here they are views of the single
    graph) count as contiguous when each row is - code that hands them on unchanged must keep working."""
    for n, t in named.items():
        if t.is_contiguous():
            continue
        if t.dim() >= 1 and t.stride(0) == 0 and t[0].is_contiguous():
            continue
        raise RuntimeError("%s must be contiguous" % n)


# The reference's loss.py calls pychain_C on EVERY training step with the same graph tensors.  Compiling the
# denominator plan (seconds for a C3-size graph) or re-uploading the numerator graphs per call would make
# that pattern ~1000x slower than the kernels, so both are memoised on what identifies the inputs:
# (data_ptr, _version, shape) of every tensor + pdf count + device.  A few entries, least recently used out.
_compat_cache = {}
_COMPAT_CACHE_ENTRIES = 8


def _tensors_key(tensors, extra):
    return tuple((t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride())) for t in tensors) + tuple(extra)


def _compat_cached(kind, tensors, extra, build):
    key = (kind,) + _tensors_key(tensors, extra)
    hit = _compat_cache.pop(key, None)
    if hit is None:
        hit = (build(), list(tensors))        # the tensors are kept alive: their data_ptr cannot be recycled
    _compat_cache[key] = hit                  # (re-)inserted last = most recently used
    while len(_compat_cache) > _COMPAT_CACHE_ENTRIES:
        _compat_cache.pop(next(iter(_compat_cache)))
    return hit[0]


class _RawGraphs(object):
    """The batched graph tensors of a pychain_C-style call as the object cpu_forward_backward reads."""
    shared_graph = None

    def __init__(self, named, log_domain):
        self.__dict__.update(named)
        self.log_domain = log_domain


def forward_backward(forward_transitions, forward_transition_indices, forward_transition_probs,
                     backward_transitions, backward_transition_indices, backward_transition_probs,
                     leaky_probs, initial_probs, final_probs, start_state, exp_nnet_output,
                     batch_sizes, sequence_lengths, num_states, leaky_hmm_coefficient=1.0e-05):
    """Same contract as `pychain_C.forward_backward`: pre-exponentiated input, returns
    [objf (0-dim), nnet_output_grad [B,T,D], ok (bool[1])]."""
    _check_contiguous(forward_transitions=forward_transitions, forward_transition_indices=forward_transition_indices,
                      forward_transition_probs=forward_transition_probs, backward_transitions=backward_transitions,
                      backward_transition_indices=backward_transition_indices,
                      backward_transition_probs=backward_transition_probs, leaky_probs=leaky_probs,
                      exp_nnet_output=exp_nnet_output, batch_sizes=batch_sizes, sequence_lengths=sequence_lengths,
                      initial_probs=initial_probs, final_probs=final_probs, start_state=start_state)
    names = _GRAPH6 + ["leaky_probs", "initial_probs", "final_probs"]
    vals = [forward_transitions, forward_transition_indices, forward_transition_probs,
            backward_transitions, backward_transition_indices, backward_transition_probs,
            leaky_probs, initial_probs, final_probs]
    if not exp_nnet_output.is_cuda:              # the reference's CPU path (chain-computation.cc:136-175,272-310): the host twin
        gb = _RawGraphs(dict(zip(names, vals)), log_domain=False)
        objf, grad, bad = cpu_forward_backward(gb, exp_nnet_output, sequence_lengths, leaky_hmm_coefficient, input_is_exp=True)
        return [objf.sum(), grad, bad == 0]
    D = exp_nnet_output.shape[2]
    dev = exp_nnet_output.device
    plan = _compat_cached("den", vals, (D, str(dev)), lambda: _plan.batch_plans(dict(zip(names, vals)), D, dev))
    objf, grad, bad = den_forward_backward(plan, exp_nnet_output, sequence_lengths,
                                           leaky_hmm_coefficient, input_is_exp=True)
    return [objf.sum(), grad, bad == 0]


def forward_backward_log_domain(forward_transitions, forward_transition_indices, forward_transition_probs,
                                backward_transitions, backward_transition_indices, backward_transition_probs,
                                initial_probs, final_probs, start_state, nnet_output,
                                batch_sizes, sequence_lengths, num_states):
    """Same contract as `pychain_C.forward_backward_log_domain`: returns
    [objf, log-grad [B,T,D] (-inf where zero), ok]."""
    _check_contiguous(forward_transitions=forward_transitions, forward_transition_indices=forward_transition_indices,
                      forward_transition_probs=forward_transition_probs, backward_transitions=backward_transitions,
                      backward_transition_indices=backward_transition_indices,
                      backward_transition_probs=backward_transition_probs, nnet_output=nnet_output,
                      batch_sizes=batch_sizes, sequence_lengths=sequence_lengths, initial_probs=initial_probs,
                      final_probs=final_probs, start_state=start_state)
    dev = nnet_output.device
    vals = [forward_transitions, forward_transition_indices, forward_transition_probs,
            backward_transitions, backward_transition_indices, backward_transition_probs,
            initial_probs, final_probs]
    if not nnet_output.is_cuda:                   # chain-log-domain-computation.cc:123-159,231-271: the host twin
        gb = _RawGraphs(dict(zip(_GRAPH6 + ["initial_probs", "final_probs"], vals)), log_domain=True)
        objf, lgrad, bad = cpu_forward_backward(gb, nnet_output, sequence_lengths, grad_mode=_lib.GRAD_LOG, clamp=False)
        return [objf.sum(), lgrad, bad == 0]
    gt = _compat_cached("num", vals, (str(dev),), lambda: {
        n: t.contiguous().to(dev) for n, t in zip(_GRAPH6 + ["initial_probs", "final_probs"], vals)})
    objf, lgrad, bad = num_forward_backward(gt, 1, num_states, nnet_output, sequence_lengths,
                                            grad_mode=_lib.GRAD_LOG)
    return [objf.sum(), lgrad, bad == 0]


def set_verbose_level(level):
    _lib.lib().pychain_hip_set_verbose_level(int(level))

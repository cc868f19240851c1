"""Hidden state and argument handling of the HIP path: two calls in flight on two streams of one
device, lengths that only the device knows, the per-step pattern of code written against pychain_C
(pychain/loss.py:44-76 calls it every step with the same graph tensors), the reference's
CHECK_CONTIGUOUS (pychain.cc:24,42-54,95-107)."""
import numpy as np
import pytest
import torch

from pychain_amd import ChainFunction, ChainGraphBatch, ChainLoss, _lib, _plan, native, synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_two_calls_in_flight_on_two_streams():
    """The side streams / events and the cached workspace are per (device, caller's stream): two fused
    ChainLoss steps issued back to back on two streams give what they give one after the other."""
    cfg = syn.CONFIGS["C3"]
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    L = [torch.tensor([400, 390, 300, 120]), torch.tensor([410, 200, 199, 3])]
    xs = [syn.make_input(4, int(l.max()), cfg["D"], seed=51 + i, device=DEV) for i, l in enumerate(L)]
    nums = [syn.make_num_graphs(l.tolist(), cfg["D"], seed=600 + 10 * i) for i, l in enumerate(L)]
    crit = ChainLoss(den, 1e-5)

    def step(i):
        xx = xs[i].clone().requires_grad_(True)
        loss = crit(xx, L[i], nums[i])
        loss.backward()
        return loss.detach(), xx.grad, ChainFunction.last_bad_count
    ref = [step(0), step(1)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rep in range(3):
        outs = [None, None]
        for i in (0, 1):
            streams[i].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(streams[i]):
                outs[i] = step(i)
        for s in streams:
            torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        for i in (0, 1):
            assert torch.equal(outs[i][0], ref[i][0]) and torch.equal(outs[i][1], ref[i][1]), (rep, i)
            assert int(outs[i][2].sum()) == 0


def test_device_lengths_out_of_range_are_clamped_and_flagged():
    """Lengths on the device are not validated on the host (no sync): a value outside [1, T] is clamped
    by the kernels - no out-of-bounds access - and reported through bad_count."""
    w = syn.make_workload("C1")
    x = w["x"].to(DEV)
    T = x.size(1)
    for bad_len in (T + 7, 0, -3):
        Ld = torch.tensor([T, bad_len], device=DEV)
        clamped = torch.tensor([T, min(max(bad_len, 1), T)])
        for graphs in (ChainGraphBatch(w["den_graph"], 2), w["num_graphs"]):
            xx = x.clone().requires_grad_(True)
            ChainFunction.apply(xx, Ld, graphs, 1e-5).backward()
            bad = int(ChainFunction.last_bad_count.sum().item())
            xr = x.clone().requires_grad_(True)
            ChainFunction.apply(xr, clamped, graphs, 1e-5).backward()
            assert bad > 0
            assert torch.equal(xx.grad[0], xr.grad[0])             # the healthy sequence is untouched
        with pytest.raises(ValueError):                            # host lengths ARE validated
            ChainFunction.apply(x.clone().requires_grad_(True), torch.tensor([T, bad_len]),
                                ChainGraphBatch(w["den_graph"], 2), 1e-5)


def _raw_args(w, den_b, xin):
    bs = torch.nn.utils.rnn.pack_padded_sequence(w["x"], w["lengths"], batch_first=True).batch_sizes
    return [den_b.forward_transitions, den_b.forward_transition_indices, den_b.forward_transition_probs,
            den_b.backward_transitions, den_b.backward_transition_indices, den_b.backward_transition_probs,
            den_b.leaky_probs, den_b.initial_probs, den_b.final_probs, den_b.start_state, xin, bs,
            w["lengths"], den_b.num_states, 1e-5]


def test_pychain_C_surface_compiles_a_graph_once(monkeypatch):
    """Code written against the reference calls pychain_C.forward_backward on every step with the same
    graph tensors: the plan is compiled (and the numerator graphs are uploaded) on the first call only;
    an in-place edit of a graph tensor is seen."""
    w = syn.make_workload("C1")
    den_b = ChainGraphBatch(w["den_graph"], 2)            # stride-0 views: accepted like .repeat copies
    x = w["x"].to(DEV).clamp(-30, 30).exp()
    native.release_workspaces()
    calls = []
    real = _plan.batch_plans
    monkeypatch.setattr(_plan, "batch_plans", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    a = native.forward_backward(*_raw_args(w, den_b, x))
    b = native.forward_backward(*_raw_args(w, den_b, x))
    assert len(calls) == 1
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and bool(a[2].all())
    w["den_graph"].final_probs.mul_(0.5)                   # the views share its storage
    c = native.forward_backward(*_raw_args(w, den_b, x))
    assert len(calls) == 2 and not torch.equal(a[0], c[0])
    w["den_graph"].final_probs.mul_(2.0)
    # numerator graphs: uploaded once
    nb = w["num_graphs"]
    bs = torch.nn.utils.rnn.pack_padded_sequence(w["x"], w["lengths"], batch_first=True).batch_sizes
    nargs = [nb.forward_transitions, nb.forward_transition_indices, nb.forward_transition_probs,
             nb.backward_transitions, nb.backward_transition_indices, nb.backward_transition_probs,
             nb.initial_probs, nb.final_probs, nb.start_state, w["x"].to(DEV).clamp(-30, 30), bs, w["lengths"],
             nb.num_states]
    n0 = len(native._compat_cache)
    r1 = native.forward_backward_log_domain(*nargs)
    r2 = native.forward_backward_log_domain(*nargs)
    assert len(native._compat_cache) == n0 + 1 and torch.equal(r1[1], r2[1])


def test_check_contiguous_like_the_reference():
    w = syn.make_workload("C1")
    den_b = ChainGraphBatch(w["den_graph"], 2)
    x = w["x"].to(DEV).clamp(-30, 30).exp()
    args = _raw_args(w, den_b, x)
    names = ["forward_transitions", "forward_transition_indices", "forward_transition_probs",
             "backward_transitions", "backward_transition_indices", "backward_transition_probs",
             "leaky_probs", "initial_probs", "final_probs", "start_state", "exp_nnet_output",
             "batch_sizes", "sequence_lengths"]
    for i, name in enumerate(names):
        t = args[i]
        if t.dim() == 1:
            nc = torch.stack([t, t], dim=1)[:, 0]                          # stride 2
        else:
            nc = t.contiguous().transpose(-1, -2).contiguous().transpose(-1, -2) if t.size(-1) > 1 and t.size(-2) > 1 else None
        if nc is None or nc.is_contiguous():
            continue
        bad_args = list(args)
        bad_args[i] = nc
        with pytest.raises(RuntimeError, match=name + " must be contiguous"):
            native.forward_backward(*bad_args)


def test_loss_total_is_the_loss_arithmetic_of_the_reference():
    """pychain_hip_loss_total = (sum den - sum num) * scale [/ norm]: -(num - den) and the division by the frame count of
    pychain/loss.py:100-104 in one launch; against torch in fp64, with and without a numerator, a device-side norm, B = 1
    and a batch larger than the kernel's block."""
    g = torch.Generator().manual_seed(3)
    for B in (1, 7, 64, 1000):
        den = (torch.randn(B, generator=g) * 300 - 4000).to(DEV)
        num = (torch.randn(B, generator=g) * 300 - 4100).to(DEV)
        want = float(den.double().sum() - num.double().sum())
        got = float(native.loss_total(den, num))
        assert abs(got - want) <= 2e-7 * abs(want) + 1e-6
        norm = torch.tensor(1234.0, device=DEV)
        got = float(native.loss_total(den, num, 0.5, norm))
        assert abs(got - want * 0.5 / 1234.0) <= 2e-7 * abs(want * 0.5 / 1234.0) + 1e-9
        got = float(native.loss_total(den, None, 2.0))
        want = 2.0 * float(den.double().sum())
        assert abs(got - want) <= 2e-7 * abs(want)


def test_step_totals_come_with_the_call():
    """The scalars of a step - the loss of pychain/loss.py:100-104, the frame count, the bad count - are written by the
    LAST workgroup of the call's last kernel (include/pychain_hip.h: totals) instead of by a chain of launch-bound scalar
    kernels of the host framework: fused ChainLoss (host and device lengths, avg on / off), the denominator ChainFunction,
    a batch larger than den_finish_kernel's block, a failing `ok`."""
    w = syn.make_workload("C1")
    x = w["x"].to(DEV)
    for avg in (True, False):
        grads = []
        for lengths in (w["lengths"], w["lengths"].to(DEV)):
            xx = x.clone().requires_grad_(True)
            loss = ChainLoss(w["den_graph"], 1e-5, avg=avg)(xx, lengths, w["num_graphs"])
            t = ChainFunction.last_totals
            assert t is not None and t.shape == (4,) and float(t[0]) == float(loss)
            assert float(t[1]) == float(w["lengths"].sum()) and float(t[2]) == 0.0
            ref = ChainLoss(w["den_graph"], 1e-5, avg=avg)
            ref.fused = False
            want = float(ref(x, w["lengths"], w["num_graphs"]))
            assert abs(float(loss) - want) <= 1e-5 * abs(want)
            raw = float(t[3])
            assert abs(raw - want * (float(w["lengths"].sum()) if avg else 1.0)) <= 1e-5 * abs(raw)
            loss.backward()
            assert torch.isfinite(xx.grad).all()
            grads.append(xx.grad.clone())
        # lengths on the device: avg's 1 / frames is divided into the gradient by the call itself, read on the device
        # (include/pychain_hip.h: loss_norm_dev) - the same gradient as with the host-side scalar, to one rounding
        assert (grads[0] - grads[1]).abs().max() <= 2e-7 * grads[0].abs().max()
        xr = x.clone().requires_grad_(True)
        ref(xr, w["lengths"], w["num_graphs"]).backward()
        assert (grads[1] - xr.grad).abs().max() <= 1e-5 * xr.grad.abs().max()
    # denominator only, B = 300 > 256 threads of the finishing workgroup; unequal lengths
    den = syn.make_den_graph(20, 60, 40, seed=0)
    B = 300
    L = torch.randint(1, 21, (B,), generator=torch.Generator().manual_seed(1))
    xb = syn.make_input(B, 20, 40, seed=3, device=DEV)
    per_seq, _, _ = native.den_forward_backward(_plan.graph_plan(den, 40, torch.device(DEV)), xb, L, 1e-5)
    o = ChainFunction.apply(xb, L, ChainGraphBatch(den, B), 1e-5)
    t = ChainFunction.last_totals
    assert abs(float(o) - float(per_seq.double().sum())) <= 1e-6 * abs(float(o)) and float(t[0]) == float(o)
    assert float(t[1]) == float(L.sum()) and float(t[2]) == 0.0
    # a NaN network output: the bad count rides in the totals
    xn = x.clone()
    xn[1, 3, 5] = float("nan")
    ChainLoss(w["den_graph"], 1e-5)(xn, w["lengths"], w["num_graphs"])
    t = ChainFunction.last_totals
    assert float(t[2]) >= 1.0 and float(t[2]) == float(ChainFunction.last_bad_count.sum())


def test_unknown_option_is_an_error():
    with pytest.raises(_lib.PychainHipError, match="unknown option"):
        with _lib.option("no_such_option"):
            pass


def test_batch_is_staged_with_one_copy_and_reordered_on_the_device():
    """A numerator batch built from a list reaches the device as ONE buffer (one H2D copy of the pinned staging buffer),
    and `reorder` re-gathers the staged copy on the device (pychain_hip_batch_reorder_dev) instead of dropping it; an
    in-place edit of a host tensor re-stages.  Reference: pychain/graph.py:122-194."""
    from pychain_amd.graph import _PACKED
    L = [50, 37, 44, 21]
    torch.cuda.init()        # (pinned staging only in a process that already talks to the GPU - never in a DataLoader worker)
    gb = syn.make_num_graphs(L, 40, seed=100, max_states=12)
    assert gb._staging is not None and gb._staging.is_pinned()
    dev = torch.device(DEV)
    dt = gb.device_tensors(dev)
    base = dt["_buffer"]
    for name, _s, _d in _PACKED[:-1]:
        if getattr(gb, name) is not None:
            assert torch.equal(dt[name].cpu(), getattr(gb, name)), name
            assert base.data_ptr() <= dt[name].data_ptr() < base.data_ptr() + base.numel()
    assert gb.device_tensors(dev) is dt                                   # cached
    order = torch.tensor([2, 0, 3])
    gb.reorder(order)                                                     # a subset: sharding does this
    dt2 = gb.device_tensors(dev)
    assert dt2 is not dt and gb.batch_size == 3
    torch.cuda.synchronize()
    for name, _s, _d in _PACKED[:-1]:
        if getattr(gb, name) is not None:
            assert torch.equal(dt2[name].cpu(), getattr(gb, name)), name
    gb.final_probs[0, 0] = -1.0                                           # in-place edit: staged copy is stale
    dt3 = gb.device_tensors(dev)
    assert dt3 is not dt2 and float(dt3["final_probs"][0, 0]) == -1.0
    # and the loss computed from the re-gathered batch is the loss of the re-gathered utterances
    x = syn.make_input(4, 50, 40, seed=5, device=DEV)
    gb2 = syn.make_num_graphs(L, 40, seed=100, max_states=12)
    gb2.device_tensors(dev)
    gb2.reorder(order)
    got = ChainFunction.apply(x[order.to(DEV)].contiguous(), torch.tensor(L)[order], gb2)
    # ... against the same graphs re-indexed on the host, tensor by tensor, and uploaded afresh
    gb3 = syn.make_num_graphs(L, 40, seed=100, max_states=12)
    with torch.no_grad():
        host = ChainGraphBatch.__new__(ChainGraphBatch)
        host.__dict__.update(gb3.__dict__)
        host._device_cache = {}
        host._staging = None
        for n in ("forward_transitions", "forward_transition_indices", "forward_transition_probs", "backward_transitions",
                  "backward_transition_indices", "backward_transition_probs", "final_probs", "initial_probs", "start_state"):
            setattr(host, n, getattr(gb3, n).index_select(0, order))
        host.batch_size = 3
    want = ChainFunction.apply(x[order.to(DEV)].contiguous(), torch.tensor(L)[order], host)
    assert float(got) == float(want)


def test_streamed_and_gated_schedules_with_a_second_process_on_the_gpu():
    """The persistent occupancy launch and the gate kernels spin on counters the recursion workgroups advance: another
    process keeping the same GPU busy (compute-bound kernels with many workgroups, as a second job, a profiler or a
    debugger would) must only slow them down - same bits, ok, no time-out into `bad`."""
    import os
    import subprocess
    import sys
    import time
    cfg = syn.CONFIGS["C3"]
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    L = torch.tensor([600, 590, 400, 333, 150, 7])
    x = syn.make_input(6, 600, cfg["D"], seed=91, device=DEV)
    numg = syn.make_num_graphs(L.tolist(), cfg["D"], seed=900)
    crit = ChainLoss(den, 1e-5)

    def step():
        xx = x.clone().requires_grad_(True)
        loss = crit(xx, L, numg)
        loss.backward()
        torch.cuda.synchronize()
        return loss.detach().clone(), xx.grad, ChainFunction.last_bad_count.tolist()
    quiet = step()
    hog = subprocess.Popen([sys.executable, "-c", (
        "import torch, time\n"
        "a = torch.randn(8192, 8192, device='cuda:0'); b = torch.randn(8192, 8192, device='cuda:0')\n"
        "torch.cuda.synchronize(); print('up', flush=True)\n"
        "t0 = time.time()\n"
        "while time.time() - t0 < 12:\n"
        "    for _ in range(20): c = a @ b\n"
        "    torch.cuda.synchronize()\n")], stdout=subprocess.PIPE, text=True,
        env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    try:
        assert hog.stdout.readline().strip() == "up"
        t0 = time.time()
        n = 0
        while time.time() - t0 < 6 and hog.poll() is None:
            for opts in ({}, {"den_segments": 3}):
                ctx = [_lib.option(k, v) for k, v in opts.items()]
                for c in ctx:
                    c.__enter__()
                try:
                    got = step()
                finally:
                    for c in reversed(ctx):
                        c.__exit__()
                assert got[2] == [0, 0], (opts, got[2])
                assert torch.equal(got[0], quiet[0]) and torch.equal(got[1], quiet[1]), opts
                n += 1
        assert n >= 4
    finally:
        hog.kill()                                   # (our own child, by handle)
        hog.wait()


def test_the_loss_scalar_takes_inplace_ops_like_the_reference():
    """the reference returns a fresh tensor (`tot_log_prob.sum()`, pychain/loss.py:78,100-104), and trainers write
    `loss /= n`.  The scalar that comes with the call (totals[4]) must not be an autograd view of a buffer created inside the
    Function (in-place ops on such views raise) and must not share an element with the step's statistics."""
    w = syn.make_workload("C1")
    for fused in (True, False):
        x = w["x"].to(DEV).requires_grad_(True)
        crit = ChainLoss(w["den_graph"], 1e-5, avg=False)
        crit.fused = fused
        loss = crit(x, w["lengths"], w["num_graphs"])
        stats = None if ChainFunction.last_totals is None else ChainFunction.last_totals.clone()
        ref = float(loss.detach())
        loss /= 4.0
        loss *= 2.0
        loss.backward()
        torch.cuda.synchronize()
        assert abs(float(loss.detach()) - ref / 2.0) <= 1e-6 * abs(ref)
        if fused:
            assert stats is not None and torch.equal(stats, ChainFunction.last_totals) and float(stats[0]) == ref
        else:
            assert ChainFunction.last_totals is None      # two native calls: no totals of the step (ShardedChainLoss falls back)
        x2 = w["x"].to(DEV).requires_grad_(True)
        crit(x2, w["lengths"], w["num_graphs"]).backward()
        assert torch.allclose(x.grad, x2.grad * 0.5, rtol=1e-6, atol=0)
    # ChainFunction alone (the denominator)
    x = w["x"].to(DEV).requires_grad_(True)
    o = ChainFunction.apply(x, w["lengths"], ChainGraphBatch(w["den_graph"], 2), 1e-5)
    o += 1.0
    o.backward()
    torch.cuda.synchronize()
    assert x.grad is not None and torch.isfinite(x.grad).all()


@pytest.mark.timeout(300)
def test_recursion_grid_that_fills_the_chip_does_not_wait_for_rows_nobody_can_write():
    """with 2B >= the CU count and pairing off (rows beyond 4096 pdfs, option den_pair = 0) every CU holds a
    recursion workgroup; if those spin on rows den_exp_rows_kernel has yet to write, that launch would never become
    resident.  Such a call must exp its rows itself: it ends, and with the same numbers as option den_dma = 2."""
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    B, T, D = max(128, (cus + 1) // 2), 64, 4100
    den = syn.make_den_graph(1100, 6000, D, seed=4)          # more than 1024 states: 16-wave workgroups, one to a CU
    L = torch.full((B,), T, dtype=torch.long)
    L[1::3] = 40
    x = syn.make_input(B, T, D, seed=5, device=DEV)
    plan = _plan.graph_plan(den, D, torch.device(DEV))
    outs = []
    for opts in ({"den_pair": 0, "den_dma": 3}, {"den_pair": 0, "den_dma": 2}):     # (3: rows exp'd ahead wherever the shape allows)
        ctx = [_lib.option(k, v) for k, v in opts.items()]
        for c in ctx:
            c.__enter__()
        try:
            assert _lib.den_kernel_names(plan.slot_rows, den.num_states, D, B)[0] == "den_recursion_lazy_kernel<dma>"
            assert _lib.lib().pychain_hip_den_uses_row_buffer(plan.stride, plan.slot_rows, den.num_states, D, B, T, 0) == 0
            assert _lib.lib().pychain_hip_den_uses_row_buffer(plan.stride, plan.slot_rows, den.num_states, D, 8, T, 0) == (1 if opts["den_dma"] == 3 else 0)
            objf, grad, bad = native.den_forward_backward(plan, x, L)
            torch.cuda.synchronize()
        finally:
            for c in reversed(ctx):
                c.__exit__()
        assert int(bad.sum()) == 0
        outs.append((objf.clone(), grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.timeout(600)
def test_a_long_lived_kernel_on_another_stream_of_the_process():
    """in a DDP step the channels of the previous bucket's RCCL all-reduce hold CUs while the loss runs.
    The recursion workgroups, the gate and the streamed occupancy launch wait for one another only in ways that end when
    that kernel ends: with 32, 64 and 224 of the CUs pinned for 30 ms by a kernel on another stream of this process
    (pychain_hip_debug_occupy) the fused step gives the SAME BITS as alone, nothing gives up into `bad`, and it ends within
    the pinned time plus a few steps (no 20 s time-out anywhere).  The timeline is written to gpurun_out/."""
    import json, os, time
    cfg = syn.CONFIGS["C3"]
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    B, T = 64, 400
    L = syn.make_lengths(B, T, "ragged", seed=5)
    num = syn.make_num_graphs(L.tolist(), cfg["D"], seed=700)
    x = syn.make_input(B, T, cfg["D"], seed=71, device=DEV)
    crit = ChainLoss(den, 1e-5)
    Ld = L.to(DEV)

    def step():
        xx = x.clone().requires_grad_(True)
        loss = crit(xx, Ld, num)
        loss.backward()
        return loss.detach(), xx.grad, ChainFunction.last_bad_count
    ref = step(); ref = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); torch.cuda.synchronize()
    alone_ms = (time.perf_counter() - t0) * 1e3
    side = torch.cuda.Stream()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rows = [dict(pinned_cus=0, step_ms=round(alone_ms, 3))]
    for pinned in (32, 64, cus - 32):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _lib.check(_lib.lib().pychain_hip_debug_occupy(pinned, 30000, side.cuda_stream), "debug_occupy")
        out = step()
        torch.cuda.current_stream().synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        assert int(out[2].sum()) == 0, pinned
        assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1]), pinned
        assert ms < 30 + 25 * max(alone_ms, 1.0), (pinned, ms, alone_ms)
        rows.append(dict(pinned_cus=pinned, step_ms=round(ms, 3)))
    root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "co_resident_kernel.json"), "w") as f:
            json.dump(dict(workload="C3 graph, B=64, T<=400 ragged, fused ChainLoss step; a 30 ms kernel pins CUs on another stream",
                           cus=cus, rows=rows), f, indent=1)
    except OSError:
        pass


@pytest.mark.parametrize("dtype,T,D", [(torch.float32, 48, 1000), (torch.bfloat16, 48, 1000), (torch.float32, 47, 1001)])
def test_a_batch_larger_than_the_chip_runs_in_slices(dtype, T, D):
    """The fused loss over B >= 7/8 of the CU count (include/pychain_hip.h: pychain_hip_chain_loss_slices) runs over slices
    of the batch in the same workspaces: per-sequence results, gradient, totals and bad count are those of the one call,
    bit for bit - ragged lengths, an odd slice, per-utterance numerator graphs (their tensors are sliced too), a NaN in
    one slice only, 2-byte rows, rows of an odd number of pdfs over an odd number of frames (a slice still starts 16-byte
    aligned: slices are multiples of 8 sequences)."""
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    B = (7 * cus + 7) // 8 + 5
    den = syn.make_den_graph(200, 2000, D)
    L = syn.make_lengths(B, T, "ragged", seed=4)
    num = syn.make_num_graphs(L.tolist(), D, seed=300)
    plan = _plan.graph_plan(den, D, torch.device(DEV))
    assert _lib.lib().pychain_hip_chain_loss_slices(plan.stride, plan.slot_rows, B) == 2
    assert _lib.lib().pychain_hip_chain_loss_slices(plan.stride, plan.slot_rows, B // 2) == 1
    x = syn.make_input(B, T, D, seed=5, device=DEV).to(dtype)
    x[B - 3, 2, 7] = float("nan")                                 # (in the last slice)

    def run(slices):
        with _lib.option("chain_slices", slices):
            xx = x.clone().requires_grad_(True)
            loss = ChainLoss(den, 1e-5, avg=True)(xx, L, num)           # (L: rebound below)
            loss.backward()
            torch.cuda.synchronize()
            return (float(loss), xx.grad.clone(), ChainFunction.last_totals.clone(), ChainFunction.last_bad_count.clone(),
                    ChainLoss.last_objf_per_seq.clone() if hasattr(ChainLoss, "last_objf_per_seq") else None)
    one, two, three = run("0"), run("-1"), run("3")
    for got in (two, three):
        assert (got[0] == one[0]) or (got[0] != got[0] and one[0] != one[0])
        assert torch.equal(torch.nan_to_num(got[1].float()), torch.nan_to_num(one[1].float()))
        assert torch.equal(torch.nan_to_num(got[2]), torch.nan_to_num(one[2])) and torch.equal(got[3], one[3])
    assert int(one[3].sum()) >= 1                                  # the NaN was counted - once
    # without the NaN: finite, and the slices' loss is the loss
    x[B - 3, 2, 7] = 0.0
    one, two = run("0"), run("-1")
    assert one[0] == one[0] and two[0] == one[0] and torch.equal(two[1], one[1]) and int(two[3].sum()) == 0
    # the lengths on the device: every slice divides its gradient by the frame count of the WHOLE batch, read on the device
    Lh, L = L, L.to(DEV)
    dev_len = run("-1")
    assert abs(dev_len[0] - one[0]) <= 1e-6 * abs(one[0])
    assert (dev_len[1].float() - one[1].float()).abs().max() <= (4e-3 if dtype != torch.float32 else 2e-7) * one[1].float().abs().max()


def test_two_criteria_and_two_host_threads_keep_their_own_reports():
    """the totals / bad count of a call travel with the tensor it returned (`loss.totals`, `loss.bad_count`);
    ShardedChainLoss reads them from there.  Two criteria interleaved in one process, then two host threads each stepping its own
    criterion on its own stream, see their own numbers - bit for bit what each gets alone."""
    import threading
    from pychain_amd.parallel import ShardedChainLoss
    w = syn.make_workload("C1")
    xa = w["x"].to(DEV)
    xb = (w["x"] * 0.5 + 0.25).to(DEV)
    xs = xa.clone()
    xs[1, 2, 3] = float("nan")
    alone = {}
    for name, x in (("a", xa), ("b", xb), ("sick", xs)):
        crit = ShardedChainLoss(w["den_graph"], 1e-5, avg=True)
        xx = x.clone().requires_grad_(True)
        l = crit(xx, w["lengths"], w["num_graphs"])
        l.backward()
        alone[name] = (l.detach().clone(), xx.grad.clone(), crit.last_stats.clone())
    # interleaved: forward a, forward sick, forward b, then read everybody's report
    ca, cs, cb = (ChainLoss(w["den_graph"], 1e-5, avg=False) for _ in range(3))
    la = ca(xa.clone().requires_grad_(True), w["lengths"], w["num_graphs"])
    ls = cs(xs.clone().requires_grad_(True), w["lengths"], w["num_graphs"])
    lb = cb(xb.clone().requires_grad_(True), w["lengths"], w["num_graphs"])
    torch.cuda.synchronize()
    assert float(la.totals[0]) == float(la.detach()) and float(lb.totals[0]) == float(lb.detach())
    assert float(la.totals[2]) == 0 and float(lb.totals[2]) == 0 and float(ls.totals[2]) >= 1
    assert int(la.bad_count.sum()) == 0 and int(ls.bad_count.sum()) >= 1 and int(lb.bad_count.sum()) == 0
    assert float(la.totals[0]) != float(lb.totals[0])
    # two host threads, a criterion and a stream each
    got, errs = {}, []

    def run(name, x):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                crit = ShardedChainLoss(w["den_graph"], 1e-5, avg=True)
                for _ in range(12):
                    xx = x.clone().requires_grad_(True)
                    l = crit(xx, w["lengths"], w["num_graphs"])
                    l.backward()
                    torch.cuda.current_stream().synchronize()
                    ref = alone[name]
                    ok = torch.equal(crit.last_stats, ref[2]) and torch.equal(xx.grad, ref[1]) if name != "sick" else float(crit.last_stats[2]) >= 1
                    got.setdefault(name, []).append(bool(ok))
        except Exception as e:          # (a thread's exception would otherwise be lost)
            errs.append(repr(e))
    ts = [threading.Thread(target=run, args=(n, x)) for n, x in (("a", xa), ("sick", xs), ("b", xb))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert all(all(v) for v in got.values()) and sorted(got) == ["a", "b", "sick"], got


def test_backward_on_workspaces_a_sliced_forward_wrote_is_refused():
    """a fused forward over a batch larger than the chip that also writes the gradient runs in slices through the SAME
    workspaces - afterwards they hold the last slice's trajectories only, and `chain_loss_backward` on that state would write a
    wrong gradient for every earlier slice.  The library remembers which forward last wrote a workspace and refuses
    (EUNSUPPORTED); the same batch forwarded WITHOUT a gradient (one call, nothing sliced) takes the backward call, and gives
    the gradient the sliced forward wrote."""
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    B, T, D = (7 * cus + 7) // 8 + 3, 40, 104
    den = syn.make_den_graph(200, 2000, D)
    L = syn.make_lengths(B, T, "ragged", seed=4)
    num = syn.make_num_graphs(L.tolist(), D, seed=300)
    plan = _plan.graph_plan(den, D, torch.device(DEV))
    assert _lib.lib().pychain_hip_chain_loss_slices(plan.stride, plan.slot_rows, B) >= 2
    gt = num.device_tensors(torch.device(DEV))
    x = syn.make_input(B, T, D, seed=5, device=DEV)
    _, _, bad, st, _ = native.chain_loss_forward(plan, gt, 1, num.num_states, x, L, 1e-5, with_grad=True, half_ok=False)
    torch.cuda.synchronize()
    assert int(bad.sum()) == 0
    with pytest.raises(_lib.PychainHipError, match="slices"):
        native.chain_loss_backward(st)
    _, _, bad2, st2, _ = native.chain_loss_forward(plan, gt, 1, num.num_states, x, L, 1e-5, with_grad=False, half_ok=False)
    g2, bad3 = native.chain_loss_backward(st2)
    torch.cuda.synchronize()
    assert int(bad2.sum()) == 0 and int(bad3.sum()) == 0
    assert (g2 - st.grad).abs().max() <= 1e-6 * st.grad.abs().max()

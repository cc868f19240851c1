"""Time segments of the denominator recursions (VERDICT r4 item 1 where the CU-time accounting lets it pay:
few sequences).  With B <= 32 the chain of T dependent frames IS the step and most CUs idle: the lazy recursions of a
(sequence, direction) are cut into 2 or 4 time segments, each started `den_tburn` frames outside itself from the ordinary start
vector - a forward / backward filter forgets where it started (profiles/r05_forgetting_table.md) - its burn-in rows discarded.
The row next to every segment is compared on the device with the TRUE row the neighbouring segment stored (1e-6 as distributions);
a miss makes the call run its recursions again, unsegmented, and is reported (totals[5]).  Replaces the T-long dependent chain of
chain-computation.cc:196-207,332-342."""
import numpy as np
import pytest
import torch

import oracle as orc
from helpers import rel_err
from pychain_amd import ChainFunction, ChainGraphBatch, ChainLoss, _lib, _plan, native, synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _den(plan, x, L, **opts):
    ctx = [_lib.option(k, v) for k, v in opts.items()]
    for c in ctx:
        c.__enter__()
    try:
        objf, grad, bad, tot = native.den_forward_backward(plan, x, L, 1e-5, totals=True)
        torch.cuda.synchronize()
    finally:
        for c in reversed(ctx):
            c.__exit__()
    return objf, grad, int(bad), tot


@pytest.mark.parametrize("name,B,T,lens", [
    ("C3", 3, 1300, [1300, 1111, 400]),            # (400 < 2 burn-ins: that sequence is not cut)
    ("C3", 5, 1024, [1024, 1023, 777, 515, 512]),
    ("C4", 2, 1100, [1100, 901]),                  # rows beyond 4096 pdfs: the LzDma map
])
def test_segmented_recursions_agree_with_the_chain(name, B, T, lens):
    cfg = syn.CONFIGS[name]
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    plan = _plan.graph_plan(den, cfg["D"], torch.device(DEV))
    L = torch.tensor(lens)
    x = syn.make_input(B, T, cfg["D"], seed=21, device=DEV)
    o0, g0, b0, t0 = _den(plan, x, L, den_tseg=0)
    assert b0 == 0 and int(t0[6]) == 1
    for S in (2, 4):
        o, g, b, t = _den(plan, x, L, den_tseg=S)
        assert b == 0 and int(t[6]) == S and int(t[5]) == 0            # nothing had to be redone
        assert float((o - o0).abs().max()) <= 1e-6 * float(o0.abs().max())
        assert rel_err(g.cpu().numpy(), g0.cpu().numpy()) <= 1e-6
        o2, g2, _, _ = _den(plan, x, L, den_tseg=S)                     # and it is a deterministic schedule
        assert torch.equal(o, o2) and torch.equal(g, g2)
    # the automatic choice for this shape: four segments (few sequences, a long chain), and against the oracle
    o, g, b, t = _den(plan, x, L)
    assert int(t[6]) == 4 and int(t[5]) == 0
    if name == "C3" and B == 3:
        ro, rg = orc.chain_function(x.cpu(), L, ChainGraphBatch(den, B), 1e-5)
        assert abs(float(o.sum()) - ro) <= 1e-4 * abs(ro) and rel_err(g.cpu().numpy(), rg) <= 1e-4
    # verbose >= 1 checks EVERY frame's invariant (chain-computation.cc:345-391) - on the same cut call, with the same bits: whether a
    # call is cut is not a function of the verbose level (a debug run must compute what production computes)
    with _lib.option("verbose", 1):
        ov, gv, bv, tv = _den(plan, x, L)
    assert int(tv[6]) == 4 and int(tv[5]) == 0 and bv == 0
    assert torch.equal(ov, o) and torch.equal(gv, g)


def test_a_burn_in_that_is_too_short_is_caught_and_redone():
    """16 frames of burn-in do not forget the start: every speculated row misses, the call runs its recursions again
    unsegmented - the SAME BITS as the unsegmented call - and says how many rows missed; nothing is `bad`."""
    cfg = syn.CONFIGS["C3"]
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    plan = _plan.graph_plan(den, cfg["D"], torch.device(DEV))
    L = torch.tensor([900, 640, 100])
    x = syn.make_input(3, 900, cfg["D"], seed=23, device=DEV)
    o0, g0, b0, t0 = _den(plan, x, L, den_tseg=0)
    o, g, b, t = _den(plan, x, L, den_tseg=4, den_tburn=16)
    assert int(t[5]) >= 6 and int(t[6]) == 4 and b == 0
    assert torch.equal(o, o0) and torch.equal(g, g0)


def test_segmented_fused_loss_two_byte_rows_and_nan():
    """The fused loss with few sequences (the numerator beside the segmented recursions), bf16 rows (the 2-byte form of the
    segmented kernel), a NaN network output inside an inner segment (it must reach the loss)."""
    cfg = syn.CONFIGS["C3"]
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    L = torch.tensor([1200, 1100, 800, 600])
    num = syn.make_num_graphs(L.tolist(), cfg["D"], seed=77)
    x = syn.make_input(4, 1200, cfg["D"], seed=25, device=DEV)

    def step(xin, **opts):
        ctx = [_lib.option(k, v) for k, v in opts.items()]
        for c in ctx:
            c.__enter__()
        try:
            xx = xin.clone().requires_grad_(True)
            loss = ChainLoss(den, 1e-5)(xx, L, num)
            loss.backward()
            torch.cuda.synchronize()
        finally:
            for c in reversed(ctx):
                c.__exit__()
        return float(loss.detach()), xx.grad, ChainFunction.last_bad_count.clone(), ChainFunction.last_totals.clone()
    l0, g0, b0, t0 = step(x, den_tseg=0)
    l1, g1, b1, t1 = step(x)                                   # automatic: 4 sequences -> segments
    assert int(b1.sum()) == 0 and abs(l1 - l0) <= 1e-6 * abs(l0) and rel_err(g1.cpu().numpy(), g0.cpu().numpy()) <= 1e-6
    ro, rg = orc.chain_loss(x.cpu(), L, den, num, 1e-5, avg=True, flavour="f64")
    assert abs(l1 - float(ro)) <= 1e-4 * abs(float(ro)) and rel_err(g1.cpu().numpy(), rg) <= 1e-5
    xh = x.to(torch.bfloat16)
    lh0, gh0, _, _ = step(xh, den_tseg=0)
    lh1, gh1, bh1, _ = step(xh, den_tseg=4)
    assert int(bh1.sum()) == 0 and abs(lh1 - lh0) <= 1e-6 * abs(lh0)
    assert rel_err(gh1.float().cpu().numpy(), gh0.float().cpu().numpy()) <= 2.0 ** -8
    # frame 700 of 1200: only an INNER alpha segment of sequence 0 stages it (the last one starts its burn-in at 708);
    # frame 100: the first segment; frame 1150: the last one; frame 650 of sequence 1 (1100 frames)
    for (b, t) in ((0, 700), (0, 100), (0, 1150), (1, 650)):
        xn = x.clone()
        xn[b, t, 11] = float("nan")
        ln, gn, bn, _ = step(xn, den_tseg=4)
        assert np.isnan(ln) and int(bn.sum()) > 0, (b, t)


def test_a_burn_in_that_misses_is_noticed_by_the_plans_controller():
    """How long a recursion needs to forget its start depends on the DATA (network outputs N(0,1) x 4 instead of x 2 need more
    than 256 frames on the benchmark graph: profiles/r05_time_segments.txt), and a call whose speculated rows do not verify runs
    its recursions twice.  Since ABI 16 the controller is IN THE LIBRARY (include/pychain_hip.h: pychain_hip_den_tseg_state): a
    device-resident state attached to the plan, read by the kernels of a segmented call and updated by the call's last kernel
    in stream order - no host read, no host timing, so the SAME sequence of calls takes the SAME decisions.
    Started here from a burn-in of 57 frames: 57, 86, 129 miss, 194 verifies; every call - also the missing ones, through the
    fallback - gives the unsegmented call's numbers; a second run from the same state repeats the first bit for bit."""
    cfg = syn.CONFIGS["C3"]
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    plan = _plan.graph_plan(den, cfg["D"], torch.device(DEV))
    B, T = 4, 1500
    L = torch.full((B,), T)
    x = syn.make_input(B, T, cfg["D"], seed=31, device=DEV)
    o0, g0, _, _ = _den(plan, x, L, den_tseg=0)
    st = plan.tseg_state
    assert st is not None and st.numel() * 4 == _lib.lib().pychain_hip_den_tseg_state_bytes()
    MAGIC = 0x74736567

    def run():
        st.zero_()
        st[0], st[1] = MAGIC, 57                      # (a state that has learnt nothing useful: burn-in 57)
        seen, grads = [], []
        for i in range(6):
            o, g, b, t = native.den_forward_backward(plan, x, L, 1e-5, totals=True)
            torch.cuda.synchronize()
            seen.append((int(t[6]), int(t[5]), int(st[1]), int(st[4])))      # segments, rows missed, burn-in AFTER the call, misses so far
            grads.append(g.clone())
            assert int(b) == 0 and rel_err(g.cpu().numpy(), g0.cpu().numpy()) <= 1e-5 and float((o - o0).abs().max()) <= 1e-5 * float(o0.abs().max())
        return seen, grads
    seen, grads = run()
    assert seen[0][0] == 4 and seen[0][1] > 0 and seen[0][2] == 86, seen                # cut, missed, redone; the next call burns in longer
    assert [s[2] for s in seen[:4]] == [86, 129, 194, 194] and seen[-1][1] == 0 and seen[-1][3] == 3, seen   # 57 -> 86 -> 129 -> 194: verifies
    again, grads2 = run()
    assert again == seen and all(torch.equal(a, b) for a, b in zip(grads, grads2))      # deterministic: no host timing in the loop
    # a caller that pins the burn-in with an option bypasses the state
    st.zero_(); st[0], st[1] = MAGIC, 57
    o, g, b, t = _den(plan, x, L, den_tburn=192)
    assert int(st[3]) == 0 and int(t[5]) == 0                                           # the state was not touched; 192 verifies


def test_a_plan_that_keeps_missing_cools_down_and_c_callers_get_the_same():
    """Beyond a third of the sequence the cut no longer pays: the plan is not cut for the next 500 calls (the segmented launch
    leaves at once, the uncut launch behind it does the work: totals[6] == 1) and starts over.  Data that forgets slowly -
    N(0,1) x 4 - at T = 640: 192 misses, 288 would not fit three times -> cool-down; misses stop after the first call."""
    cfg = syn.CONFIGS["C3"]
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    plan = _plan.graph_plan(den, cfg["D"], torch.device(DEV))
    B, T = 4, 640
    L = torch.full((B,), T)
    x = syn.make_input(B, T, cfg["D"], seed=33, device=DEV) * 2.0            # (make_input is N(0,1) x 2)
    o0, g0, _, _ = _den(plan, x, L, den_tseg=0)
    st = plan.tseg_state
    st.zero_()
    seen = []
    for i in range(4):
        o, g, b, t = native.den_forward_backward(plan, x, L, 1e-5, totals=True)
        torch.cuda.synchronize()
        seen.append((int(t[6]), int(t[5]), int(st[1]), int(st[2]), int(st[3])))
        assert int(b) == 0 and rel_err(g.cpu().numpy(), g0.cpu().numpy()) <= 1e-5
    assert seen[0][0] > 1 and seen[0][1] > 0 and seen[0][3] == 500, seen     # cut, missed, cooling down from here on
    assert all(s[0] == 1 and s[1] == 0 for s in seen[1:]) and seen[-1][3] == 497 and seen[-1][4] == 4, seen
    assert torch.equal(g, g0)                                                # a cooled-down call IS the uncut call
    st.zero_()

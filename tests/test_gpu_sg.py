"""Graphs whose arcs carry the pdf of the state they ENTER ("pdf by state": a chain denominator compiled from a phone LM over
HMM topologies - VERDICT r4 / r5: "exploitation of the structure real denominators have").  The plan compiler marks such a
graph (plan_format.h: PLAN_FLAG_PDF_BY_STATE, launch-hint bit 27) and the lazy recursions run their one-gather form
(den_lazy.inc.h: SG): alpha multiplies a row's sum by the nnet output once, where the row's value is formed; beta gathers from a
vector that was pre-multiplied where it was written, its nnet-output rows running a frame ahead.  Replaces the per-arc gather of
chain-computation.cc:150-174 / :289-309 (`x[s,t,pdf_k]` read once per transition).  Held to the fp64 oracle and to the ordinary
kernels on the SAME plan (option den_sg = 0)."""
import numpy as np
import pytest
import torch

import oracle as orc
from helpers import record_parity, rel_err
from pychain_amd import ChainFunction, ChainGraph, ChainGraphBatch, ChainLoss, _lib, _plan, native, synthetic as syn
from pychain_amd.simplefst import StdVectorFst

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D = 3456


@pytest.fixture(scope="module")
def den():
    return syn.make_structured_den_graph()


def _call(den, x, L, **opts):
    xx = x.clone().requires_grad_(True)
    ctx = [_lib.option(k, v) for k, v in opts.items()]
    for c in ctx:
        c.__enter__()
    try:
        o = ChainFunction.apply(xx, L, ChainGraphBatch(den, x.size(0)), 1e-5)
        o.backward()
        torch.cuda.synchronize()
    finally:
        for c in reversed(ctx):
            c.__exit__()
    return float(o.detach()), xx.grad, int(o.bad_count.sum()), o.totals_all


def _name(den, B, fused=False, **opts):
    plan = _plan.graph_plan(den, D, torch.device(DEV))
    ctx = [_lib.option(k, v) for k, v in opts.items()]
    for c in ctx:
        c.__enter__()
    try:
        return _lib.den_kernel_names(plan.slot_rows, plan.num_states, D, B, fused=fused)[0]
    finally:
        for c in reversed(ctx):
            c.__exit__()


def test_the_structured_graph_runs_the_one_gather_recursions(den):
    plan = _plan.graph_plan(den, D, torch.device(DEV))
    assert (plan.slot_rows >> 27) & 1                                       # marked by the plan compiler
    assert "one gather per arc" in _name(den, 8) and "one gather per arc" in _name(den, 64, fused=True)
    assert "one gather per arc" not in _name(den, 8, den_sg=0)
    assert "one gather per arc" not in _name(den, 8, den_dma=0)             # rows through registers: the ordinary kernels
    rnd = syn.make_den_graph(3000, 30000, D, seed=0)                        # the benchmark graph: every arc its own pdf
    assert not (_plan.graph_plan(rnd, D, torch.device(DEV)).slot_rows >> 27) & 1


@pytest.mark.parametrize("lens", [[700, 651, 512, 333, 2, 1], [257, 256, 255, 3], [1], [2, 2]])
def test_one_gather_recursions_vs_oracle_and_vs_the_ordinary_kernels(den, lens):
    """Ragged lengths down to one and two frames (beta's row pipeline runs a frame ahead: the short ends are its edge cases),
    odd and even; objective and gradient against the fp64 oracle (1e-5) and against the ordinary recursions on the same plan."""
    T, B = max(lens), len(lens)
    L = torch.tensor(lens)
    x = syn.make_input(B, T, D, seed=90 + T, device=DEV)
    o, g, bad, _ = _call(den, x, L)
    o0, g0, bad0, _ = _call(den, x, L, den_sg=0)
    ro, rg = orc.chain_function(x.cpu(), L, ChainGraphBatch(den, B), 1e-5, flavour="f64")
    e, e0 = rel_err(g.cpu().numpy(), rg), rel_err(g.cpu().numpy(), g0.cpu().numpy())
    record_parity("pdf_by_state_T%d_B%d" % (T, B), objf=abs(o - ro) / abs(ro), grad_vs_f64=e, grad_vs_ordinary_kernels=e0,
                  ordinary_vs_f64=rel_err(g0.cpu().numpy(), rg), bound=1e-5)
    assert bad == 0 and bad0 == 0
    assert abs(o - ro) <= 1e-5 * abs(ro) and e <= 1e-5, (o, ro, e)
    assert abs(o - o0) <= 1e-5 * abs(o0) and e0 <= 1e-5, (o, o0, e0)
    # deterministic: the same call gives the same bits
    o2, g2, _, _ = _call(den, x, L)
    assert o2 == o and torch.equal(g2, g)


def test_one_gather_recursions_in_time_segments_and_in_the_fused_loss(den):
    """The same form cut into time segments (few sequences: DESIGN.md §3.13) - verified splices, nothing redone - and inside the
    fused ChainLoss with per-utterance numerators, against the oracle."""
    L = torch.tensor([900, 820, 400])
    x = syn.make_input(3, 900, D, seed=17, device=DEV)
    o, g, bad, tot = _call(den, x, L, den_tseg=2, den_tburn=192)
    o0, g0, bad0, _ = _call(den, x, L, den_tseg=0)
    assert bad == 0 and bad0 == 0 and int(tot[6]) == 2 and int(tot[5]) == 0
    assert abs(o - o0) <= 1e-6 * abs(o0) and rel_err(g.cpu().numpy(), g0.cpu().numpy()) <= 1e-5
    num = syn.make_num_graphs(L.tolist(), D, seed=100)
    res = {}
    for sg in (1, 0):
        with _lib.option("den_sg", sg):
            xx = x.clone().requires_grad_(True)
            loss = ChainLoss(den, 1e-5, avg=True)(xx, L, num)
            loss.backward()
            torch.cuda.synchronize()
            assert int(loss.bad_count.sum()) == 0
            res[sg] = (float(loss.detach()), xx.grad.cpu().numpy())
    rl, rg = orc.chain_loss(x.cpu(), L, den, num, 1e-5, avg=True, flavour="f64")
    assert abs(res[1][0] - rl) <= 1e-5 * abs(rl) and rel_err(res[1][1], rg) <= 1e-5
    assert rel_err(res[1][1], res[0][1]) <= 1e-5


def test_a_nan_and_values_beyond_the_clamp(den):
    """NaN network outputs turn the objective NaN and `ok` false; values beyond +-30 are clamped in-kernel as everywhere."""
    L = torch.tensor([300, 280])
    x = syn.make_input(2, 300, D, seed=5, device=DEV)
    x[0, ::7, ::11] *= 40.0
    o, g, bad, _ = _call(den, x, L)
    ro, rg = orc.chain_function(x.cpu(), L, ChainGraphBatch(den, 2), 1e-5, flavour="f64")
    assert bad == 0 and abs(o - ro) <= 1e-5 * abs(ro) and rel_err(g.cpu().numpy(), rg) <= 1e-5
    x[1, 100, 5] = float("nan")
    o, g, bad, _ = _call(den, x, L)
    assert bad >= 1 and np.isnan(o)


def test_a_graph_where_only_some_states_qualify_runs_the_ordinary_kernels():
    """ONE arc that carries another pdf than its destination's other arcs: the plan is not marked, the ordinary recursions run,
    and the result is the oracle's - per plan, no mixture inside a launch."""
    n, fan = 1500, 9
    base = syn.make_structured_den_graph(n, fan, D)
    ft = base.forward_transitions.clone()
    k = int(torch.nonzero(ft[:, 1] == 7)[0])                               # an arc entering state 7 ...
    other = (int(ft[k, 2]) + 1) % D
    arcs_src, arcs_dst, arcs_pdf = ft[:, 0].numpy().copy(), ft[:, 1].numpy().copy(), ft[:, 2].numpy().copy()
    arcs_pdf[k] = other                                                    # ... now with a pdf of its own
    lp = np.log(base.forward_transition_probs.numpy())
    fst = StdVectorFst.from_arrays(2 * n, 0, arcs_src, arcs_dst, arcs_pdf, lp, np.zeros(2 * n))
    mixed = ChainGraph(fst, initial_mode="leaky", final_mode="ones", log_domain=False)
    plan = _plan.graph_plan(mixed, D, torch.device(DEV))
    assert not (plan.slot_rows >> 27) & 1 and "one gather per arc" not in _name(mixed, 4)
    L = torch.tensor([300, 299, 150, 9])
    x = syn.make_input(4, 300, D, seed=23, device=DEV)
    o, g, bad, _ = _call(mixed, x, L)
    ro, rg = orc.chain_function(x.cpu(), L, ChainGraphBatch(mixed, 4), 1e-5, flavour="f64")
    assert bad == 0 and abs(o - ro) <= 1e-5 * abs(ro) and rel_err(g.cpu().numpy(), rg) <= 1e-5


@pytest.mark.parametrize("lens,opts", [([700, 651, 512, 333, 104, 103, 2, 1], {}), ([1500, 1400, 900, 300], {}),
                                       ([900, 820, 400], dict(den_tseg=2, den_tburn=192)), ([1500, 1211], dict(den_tseg=4, den_tburn=160))])
def test_crossing_recursions_vs_the_streamed_occupancy_launch(den, lens, opts):
    """Option den_cross = 1 (off by default - measured slower): each recursion emits the occupancies of its own
    second half - a(t+1,j) b(t+1,j) against the other side's row, landed a step ahead, into fixed-point accumulators per pdf - and the
    occupancy launch handles a band of 48 frames around every middle.  Same objective, same gradient to 1e-6 of its largest entry
    (fixed point at 2^30 / G': 3e-7 measured) and rows that sum to 1 as they do without it; sequences too short to have two halves
    (< 104 frames) are left to the occupancy launch whole; uncut and cut into time segments (a segment has its own middle)."""
    T, B = max(lens), len(lens)
    L = torch.tensor(lens)
    x = syn.make_input(B, T, D, seed=90 + T, device=DEV)
    with _lib.option("den_cross", 1):
        assert "crossing" in _name(den, B) and "crossing" not in _name(den, B, fused=True)
    assert "crossing" not in _name(den, B)
    o, g, bad, tot = _call(den, x, L, den_cross=1, **opts)
    o0, g0, bad0, tot0 = _call(den, x, L, **opts)
    assert bad == 0 and bad0 == 0 and o == o0
    assert torch.equal(tot[5:7], tot0[5:7])                                # the same segments, the same (zero) misses
    gm = float(g0.abs().max())
    err = float((g - g0).abs().max()) / gm
    rows = g.sum(dim=2)
    for b, Lb in enumerate(lens):
        assert float((rows[b, :Lb] - g0.sum(dim=2)[b, :Lb]).abs().max()) <= 2e-6 and float(g[b, Lb:].abs().max() if Lb < T else 0.0) == 0.0
    ro, rg = orc.chain_function(x.cpu(), L, ChainGraphBatch(den, B), 1e-5, flavour="f64")
    record_parity("crossing_T%d_B%d_%s" % (T, B, "tseg%d" % opts["den_tseg"] if opts else "uncut"), grad_vs_streamed_occupancies=err,
                  grad_vs_f64=rel_err(g.cpu().numpy(), rg), bound=1e-6)
    assert err <= 1e-6 and rel_err(g.cpu().numpy(), rg) <= 1e-5, err
    o2, g2, _, _ = _call(den, x, L, den_cross=1, **opts)                   # integer accumulation: the same bits every time
    assert o2 == o and torch.equal(g2, g)


def test_crossing_with_states_on_several_alpha_positions_and_the_per_frame_check():
    """A hub state with more arcs than a slot-row group holds is dealt to several ALPHA positions (plan.cpp): its beta lane pairs
    with the first, the table of further positions (PlanHeader::off_extra_a) with the rest.  And verbose = 1: the 5 % invariant of
    chain-computation.cc:363-390 is checked on every frame, on the totals the crossing measures itself."""
    n, fan = 600, 6
    base = syn.make_structured_den_graph(n, fan, D)
    ft = base.forward_transitions
    src, dst, pdf = ft[:, 0].numpy().copy(), ft[:, 1].numpy().copy(), ft[:, 2].numpy().copy()
    lp = np.log(base.forward_transition_probs.numpy())
    rng = np.random.default_rng(3)
    hub, hub_pdf = 11, int(pdf[np.nonzero(dst == 11)[0][0]])
    extra = rng.choice(2 * n, size=200, replace=False)                     # 200 more arcs ENTER the hub, all with its pdf
    src = np.concatenate([src, extra]); dst = np.concatenate([dst, np.full(200, hub)]); pdf = np.concatenate([pdf, np.full(200, hub_pdf)])
    lp = np.concatenate([lp, np.full(200, np.log(0.01))])
    fst = StdVectorFst.from_arrays(2 * n, 0, src, dst, pdf, lp, np.zeros(2 * n))
    g_hub = ChainGraph(fst, initial_mode="leaky", final_mode="ones", log_domain=False)
    plan = _plan.graph_plan(g_hub, D, torch.device(DEV))
    assert (plan.slot_rows >> 27) & 1 and plan.num_states > 2 * n          # (positions of the longer side: the hub sits on several)
    L = torch.tensor([420, 333, 209])
    x = syn.make_input(3, 420, D, seed=31, device=DEV)
    o, g, bad, _ = _call(g_hub, x, L, den_cross=1)
    o0, g0, bad0, _ = _call(g_hub, x, L)
    ro, rg = orc.chain_function(x.cpu(), L, ChainGraphBatch(g_hub, 3), 1e-5, flavour="f64")
    assert bad == 0 and bad0 == 0 and o == o0
    assert float((g - g0).abs().max()) <= 1e-6 * float(g0.abs().max()) and rel_err(g.cpu().numpy(), rg) <= 1e-5
    o1, g1, bad1, _ = _call(g_hub, x, L, den_cross=1, verbose=1)
    assert bad1 == 0 and o1 == o and torch.equal(g1, g)


def test_crossing_after_a_splice_miss_is_the_uncut_crossing(den):
    """Time segments with 16 frames of burn-in do not verify: the call runs its recursions again uncut - with the
    crossing on, the uncut crossing, whose middles are the sequences' and not the segments'; the occupancy launch must then handle
    THOSE bands.  Same bits as the uncut crossing asked for directly; the misses are reported; nothing is `bad`."""
    L = torch.tensor([900, 640, 300])
    x = syn.make_input(3, 900, D, seed=23, device=DEV)
    o0, g0, b0, t0 = _call(den, x, L, den_cross=1, den_tseg=0)
    o, g, b, t = _call(den, x, L, den_cross=1, den_tseg=4, den_tburn=16)
    assert b == 0 and b0 == 0 and int(t[6]) == 4 and int(t[5]) >= 1
    assert o == o0 and torch.equal(g, g0)
    o1, g1, b1, _ = _call(den, x, L, den_tseg=0)
    assert b1 == 0 and float((g - g1).abs().max()) <= 1e-6 * float(g1.abs().max())


def test_crossing_at_full_size_every_workgroup_waits_for_a_peer():
    """64 sequences of 1500 frames on the C3-sized pdf-by-state graph - the shape the option is for: 128 recursion workgroups (256 in
    two time segments), each landing rows of a peer from the middle on; against the streamed path on the same input, and the
    size-independent properties: rows of the gradient sum to the gradient scale, zeros in the padding, the objective bit-identical."""
    den = syn.make_structured_den_graph()
    B, T = 64, 1500
    lens = [T] * 40 + [1500 - 37 * i for i in range(1, 25)]               # 40 full-length sequences, 24 ragged ones down to 612 frames
    L = torch.tensor(lens)
    x = syn.make_input(B, T, D, seed=4, device=DEV)
    for opts in ({"den_tseg": 0}, {"den_tseg": 2}):
        o, g, bad, tot = _call(den, x, L, den_cross=1, **opts)
        o0, g0, bad0, tot0 = _call(den, x, L, **opts)
        assert bad == 0 and bad0 == 0 and o == o0 and int(tot[5]) == 0
        gm = float(g0.abs().max())
        err = float((g - g0).abs().max()) / gm
        record_parity("crossing_full_size_%s" % ("tseg2" if opts["den_tseg"] else "uncut"), grad_vs_streamed_occupancies=err, bound=1e-6)
        assert err <= 1e-6, err
        rows = g.sum(dim=2)
        for b in (0, 39, 40, 63):
            assert float((rows[b, :lens[b]] - 1.0).abs().max()) <= 2e-5 and (lens[b] == T or float(g[b, lens[b]:].abs().max()) == 0.0)
        del g, g0

"""The general denominator kernels (den_general.hip): graphs the compiled-plan kernels do not take - more than 65 535
states or pdfs, or vectors beyond the LDS of one CU - run instead of being refused; the reference's CPU path has no size
limit (chain-computation.cc:113-176,247-311).  Against the oracle, and against the fast kernels on graphs both take."""
import numpy as np
import pytest
import torch

import oracle as orc
from helpers import rel_err
from pychain_amd import ChainFunction, ChainGraphBatch, ChainLoss, _lib, _plan, synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _den(x, L, den):
    xx = x.clone().requires_grad_(True)
    o = ChainFunction.apply(xx, L, ChainGraphBatch(den, x.size(0)), 1e-5)
    o.backward()
    torch.cuda.synchronize()
    return float(o.detach()), xx.grad, int(ChainFunction.last_bad_count.sum())


def _kernel(den, D, B):
    plan = _plan.graph_plan(den, D, torch.device(DEV))
    return _lib.den_kernel_names(plan.slot_rows, den.num_states, D, B)[0]


@pytest.mark.parametrize("H,K,D", [(20, 60, 40), (200, 2000, 1000), (700, 6000, 3456)])
def test_forced_general_path_vs_the_tile_kernels_and_the_oracle(monkeypatch, H, K, D):
    L = torch.tensor([61, 61, 40, 1])
    x = syn.make_input(4, 61, D, seed=17, device=DEV)
    fast = syn.make_den_graph(H, K, D, seed=2)
    o_f, g_f, bad_f = _den(x, L, fast)
    monkeypatch.setenv("PYCHAIN_PLAN_GENERAL", "1")
    gen = syn.make_den_graph(H, K, D, seed=2)                        # (a new object: plans are cached on the graph)
    assert _kernel(gen, D, 4) == "den_general_recursion_kernel" and _kernel(fast, D, 4) != "den_general_recursion_kernel"
    o_g, g_g, bad_g = _den(x, L, gen)
    assert bad_f == 0 and bad_g == 0
    assert abs(o_g - o_f) <= 1e-5 * abs(o_f) and rel_err(g_g.cpu().numpy(), g_f.cpu().numpy()) <= 1e-5
    ro, rg = orc.chain_function(x.cpu(), L, ChainGraphBatch(gen, 4), 1e-5)
    assert abs(o_g - ro) <= 1e-4 * abs(ro) and rel_err(g_g.cpu().numpy(), rg) <= 1e-4
    assert bool((g_g[2, 40:] == 0).all()) and bool((g_g[3, 1:] == 0).all())
    # the fused loss over it (numerator accumulated afterwards: the fold needs the two-frame tile kernel)
    numg = syn.make_num_graphs(L.tolist(), D, seed=300, max_states=10)
    xx = x.clone().requires_grad_(True)
    loss = ChainLoss(gen, 1e-5)(xx, L, numg)
    loss.backward()
    rl, rgr = orc.chain_loss(x.cpu(), L, gen, numg, 1e-5, avg=True)
    assert abs(float(loss.detach()) - float(rl)) <= 1e-4 * abs(float(rl)) and rel_err(xx.grad.cpu().numpy(), rgr) <= 1e-4
    # ok semantics are shared with the fast path: a NaN network output, the 5 % invariant
    xn = x.clone()
    xn[0, 3, 1] = float("nan")
    o, g, bad = _den(xn, L, gen)
    assert bad > 0 and np.isnan(o)
    with _lib.option("debug_corrupt_row", "den,1,0,1.2"):
        o, g, bad = _den(x, L, gen)
    assert bad > 0 and o == o_g


@pytest.mark.parametrize("H,K,D", [(70000, 150000, 64), (64, 3000, 70000), (300, 2400, 40000)])
def test_shapes_beyond_the_tile_kernels_vs_oracle(H, K, D):
    """More than 65 535 states; more than 65 535 pdfs; a nnet-output row that does not fit the LDS next to the state
    vector: refused in rounds 1-2 (EUNSUPPORTED), now served by the general kernels."""
    den = syn.make_den_graph(H, K, D, seed=4)
    L = torch.tensor([13, 9])
    x = syn.make_input(2, 13, D, seed=23, device=DEV)
    assert _kernel(den, D, 2) == "den_general_recursion_kernel"
    o, g, bad = _den(x, L, den)
    assert bad == 0
    ro, rg = orc.chain_function(x.cpu(), L, ChainGraphBatch(den, 2), 1e-5)
    assert abs(o - ro) <= 1e-4 * abs(ro) and rel_err(g.cpu().numpy(), rg) <= 1e-4


# ---- numerator graphs beyond the tile kernels (num_general.hip) ----------------------------------------------------------
def _fan_fst(H, D, seed):
    """A log-domain graph of H states whose paths are short whatever H is: the start state (self-loop) fans out to every
    middle state, every middle state (self-loop on each third one) goes to the last, final state (self-loop).  In-degree
    and out-degree H - 2 at the two ends."""
    from pychain_amd.simplefst import StdVectorFst
    rs = np.random.RandomState(seed)
    pdf = lambda: int(rs.randint(D))
    arcs = [(0, 0, pdf(), -0.6)] + [(0, s, pdf(), -1.0 - (s % 7) * 0.1) for s in range(1, H - 1)]
    for s in range(1, H - 1):
        if s % 3 == 0:
            arcs.append((s, s, pdf(), -0.9))
        arcs.append((s, H - 1, pdf(), -0.4))
    arcs.append((H - 1, H - 1, pdf(), -0.5))
    return StdVectorFst.from_arcs(H, 0, arcs, {H - 1: 0.0})


@pytest.mark.parametrize("H,D,T", [(70000, 64, 12), (33, 70000, 40), (200, 40000, 33)])
def test_numerator_graphs_beyond_the_tile_kernels_vs_oracle(H, D, T):
    """More than 65 535 numerator states; more than 65 535 pdfs; a nnet-output row + graph that do not fit the LDS: refused
    in rounds 1-3 (EUNSUPPORTED), now served by num_general.hip like the reference's CPU path serves them
    (chain-log-domain-computation.cc:123-159 has no size limit).  Linear gradient through ChainFunction and the reference's
    log-gradient contract through the pychain_C surface, ragged lengths, against the oracle in fp64 (1e-5) and fp32 (1e-4);
    a NaN network output on an arc's pdf and the 5 % invariant are seen."""
    from pychain_amd import ChainGraph, native
    graphs = [ChainGraph(_fan_fst(H, D, 900 + i), log_domain=True) for i in range(2)]
    gb = ChainGraphBatch(graphs, max_num_transitions=max(g.num_transitions for g in graphs), max_num_states=H)
    L = torch.tensor([T, max(3, (2 * T) // 3)])
    # (network outputs within about a nat of each other: with a fan-in of 70 000 the reference's LogAdd, which drops terms
    # 15.9 nats below the running sum - base.h:14-32 - would otherwise differ from exact arithmetic by what it dropped)
    x = syn.make_input(2, T, D, seed=41, scale=0.3, device=DEV)
    xx = x.clone().requires_grad_(True)
    o = ChainFunction.apply(xx, L, gb)
    o.backward()
    torch.cuda.synchronize()
    assert int(ChainFunction.last_bad_count.sum()) == 0
    g = xx.grad.cpu().numpy()
    ro64, rg64 = orc.chain_function(x.cpu(), L, gb, flavour="f64")
    assert abs(float(o.detach()) - ro64) <= 1e-5 * abs(ro64) and rel_err(g, rg64) <= 1e-5, rel_err(g, rg64)
    # the fp32 restatement: within 1e-4 where its own sums are short; a fan-in of 70 000 float LogAdds is itself 1.4e-3 from
    # exact arithmetic, and the bound is then that distance + 1e-5 (triangle inequality)
    ro, rg = orc.chain_function(x.cpu(), L, gb)
    assert abs(float(o.detach()) - ro) <= 1e-4 * abs(ro) and rel_err(g, rg) <= max(1e-4, rel_err(rg, rg64) + 1e-5)
    assert bool((xx.grad[1, int(L[1]):] == 0).all())
    assert abs(float(xx.grad[0, 0].sum()) - 1.0) <= 1e-4          # a frame's occupancies sum to one
    # the reference's log-gradient contract (-inf where zero) through the pychain_C surface
    bs = torch.nn.utils.rnn.pack_padded_sequence(x.cpu(), L, batch_first=True).batch_sizes
    objf, lg, ok = native.forward_backward_log_domain(
        gb.forward_transitions, gb.forward_transition_indices, gb.forward_transition_probs, gb.backward_transitions,
        gb.backward_transition_indices, gb.backward_transition_probs, gb.initial_probs, gb.final_probs, gb.start_state,
        x.clamp(-30, 30), bs, L, gb.num_states)
    assert bool(ok) and abs(float(objf) - ro64) <= 1e-5 * abs(ro64)
    lgc = lg.cpu().numpy()
    assert np.array_equal(np.isneginf(lgc), rg64 == 0.0) or np.isneginf(lgc[rg64 > 1e-30]).sum() == 0
    big = rg64 > 1e-6
    np.testing.assert_allclose(np.exp(lgc[big]), rg64[big], rtol=1e-4)
    assert bool(np.isneginf(lgc[1, int(L[1]):]).all())
    # fused loss over a denominator of the same pdf count (its plan is general too where D is beyond the tile kernels)
    if H <= 1000:
        den = syn.make_den_graph(40, 200, D, seed=6)
        x2 = x.clone().requires_grad_(True)
        loss = ChainLoss(den, 1e-5)(x2, L, gb)
        loss.backward()
        rl, rgr = orc.chain_loss(x.cpu(), L, den, gb, 1e-5, avg=True, flavour="f64")
        assert abs(float(loss.detach()) - float(rl)) <= 1e-4 * abs(float(rl)) and rel_err(x2.grad.cpu().numpy(), rgr) <= 1e-5
        assert ChainFunction.last_bad_count.tolist() == [0, 0]
    # `ok`: a NaN on a pdf an arc emits; one stored row scaled by 20 %
    pdf0 = int(gb.forward_transitions[0, 0, 2])
    xn = x.clone()
    xn[0, 0, pdf0] = float("nan")
    on = ChainFunction.apply(xn.requires_grad_(True), L, gb)
    torch.cuda.synchronize()
    assert int(ChainFunction.last_bad_count.sum()) > 0 and np.isnan(float(on.detach()))
    with _lib.option("debug_corrupt_row", "num,1,0,1.2"):
        ChainFunction.apply(x.clone().requires_grad_(True), L, gb)
        torch.cuda.synchronize()
        assert int(ChainFunction.last_bad_count.sum()) > 0

"""The reference's `ok` flag (third return value of pychain_C.forward_backward[_log_domain],
pychain.cc:72,122) on the HIP path: bad_count == 0 on healthy input - also with the check on every
frame (verbose level >= 1, chain-computation.cc:337-338) - and > 0, without a hang or an exception,
when the computation broke down: NaN in the network output, a denominator that cannot end, a numerator
graph that does not fit its utterance (chain-computation.cc:345-391, chain-log-domain-computation.cc:283-304)."""
import numpy as np
import pytest
import torch

from pychain_amd import ChainFunction, ChainGraph, ChainGraphBatch, ChainLoss, _lib, native, synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _den(x, lengths, den, leaky=1e-5):
    xx = x.clone().requires_grad_(True)
    objf = ChainFunction.apply(xx, lengths, ChainGraphBatch(den, x.size(0)), leaky)
    objf.backward()
    torch.cuda.synchronize()
    return float(objf.detach()), xx.grad, int(ChainFunction.last_bad_count.sum().item())


def _num(x, lengths, graphs):
    xx = x.clone().requires_grad_(True)
    objf = ChainFunction.apply(xx, lengths, graphs)
    objf.backward()
    torch.cuda.synchronize()
    return float(objf.detach()), xx.grad, int(ChainFunction.last_bad_count.sum().item())


@pytest.fixture
def verbose_level():
    def set_level(n):
        native.set_verbose_level(n)
    yield set_level
    native.set_verbose_level(0)


def test_healthy_input_is_ok_with_the_check_on_every_frame(verbose_level):
    """No false alarm: C3 graph, T = 700 (long enough for rounding to accumulate in the stored scales),
    ragged lengths, both recursion kernels; verbose level 1 checks every frame and must not change a bit."""
    cfg = syn.CONFIGS["C3"]
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    L = torch.tensor([700, 640, 333, 2])
    x = syn.make_input(4, 700, cfg["D"], seed=41, device=DEV)
    numg = syn.make_num_graphs(L.tolist(), cfg["D"], seed=500)
    for lazy in (1, 0):
        _lib.lib().pychain_hip_set_den_lazy(lazy)
        try:
            # (few sequences: a call at verbose level 0 would be cut into time segments, one at level >= 1 is not - DESIGN.md
            # §3.13; what must not change a bit is the CHECK, so both sides run uncut)
            with _lib.option("den_tseg", 0):
                o0, g0, bad0 = _den(x, L, den)
                verbose_level(1)
                assert _lib.lib().pychain_hip_get_verbose_level() == 1
                o1, g1, bad1 = _den(x, L, den)
                verbose_level(0)
        finally:
            _lib.lib().pychain_hip_set_den_lazy(1)
        assert bad0 == 0 and bad1 == 0, (lazy, bad0, bad1)
        assert o0 == o1 and torch.equal(g0, g1)
    o0, g0, bad0 = _num(x, L, numg)
    verbose_level(1)
    o1, g1, bad1 = _num(x, L, numg)
    assert bad0 == 0 and bad1 == 0 and o0 == o1 and torch.equal(g0, g1)
    # the fused loss reports [denominator, numerator]
    xx = x.clone().requires_grad_(True)
    ChainLoss(den, 1e-5)(xx, L, numg).backward()
    torch.cuda.synchronize()
    assert ChainFunction.last_bad_count.tolist() == [0, 0]


def test_nan_network_output_is_not_ok():
    """torch.clamp / exp propagate NaN (loss.py:30,43): the loss is NaN and ok is false - the fused
    clamp must not turn a diverged network into a healthy-looking one."""
    w = syn.make_workload("C1")
    x = w["x"].to(DEV).clone()
    x[1, 7, 3] = float("nan")
    o, g, bad = _den(x, w["lengths"], w["den_graph"])
    assert bad > 0 and np.isnan(o)
    # the numerator only reads the pdfs on its arcs (the reference likewise): a NaN it never gathers is harmless,
    # one on the first arc of the start state at t = 0 is not
    o, g, bad = _num(x, w["lengths"], w["num_graphs"])
    assert bad == 0 and np.isfinite(o)
    x[1, 0, int(w["num_graphs"].forward_transitions[1, 0, 2])] = float("nan")
    o, g, bad = _num(x, w["lengths"], w["num_graphs"])
    assert bad > 0
    xx = x.clone().requires_grad_(True)
    loss = ChainLoss(w["den_graph"], 1e-5)(xx, w["lengths"], w["num_graphs"])
    loss.backward()
    torch.cuda.synchronize()
    assert int(ChainFunction.last_bad_count.sum().item()) > 0 and np.isnan(float(loss.detach()))
    # the other sequence is untouched by its neighbour's NaN
    assert bool(torch.isfinite(xx.grad[0]).all())


def test_infinite_network_output_is_clamped_like_the_reference():
    """clamp(+-inf) = +-30 (loss.py:30): finite results, ok."""
    w = syn.make_workload("C1")
    x = w["x"].to(DEV).clone()
    x[0, 3, 5] = float("inf")
    x[1, 9, 2] = float("-inf")
    xc = x.clone()
    xc[0, 3, 5] = 30.0
    xc[1, 9, 2] = -30.0
    o, g, bad = _den(x, w["lengths"], w["den_graph"])
    oc, gc, badc = _den(xc, w["lengths"], w["den_graph"])
    assert bad == 0 and badc == 0 and o == oc and torch.equal(g, gc)


def test_denominator_that_cannot_end_is_not_ok():
    """All final probabilities zero: sum_i alpha'(T,i) final(i) = 0, log-likelihood -inf
    (chain-computation.cc:209-230): ok is false, nothing hangs, nothing raises."""
    w = syn.make_workload("C1")
    den = w["den_graph"]
    g2 = ChainGraph.from_tensors(
        den.forward_transitions, den.forward_transition_probs, den.forward_transition_indices,
        den.backward_transitions, den.backward_transition_probs, den.backward_transition_indices,
        torch.zeros_like(den.final_probs), den.initial_probs, den.leaky_probs,
        start_state=den.start_state, log_domain=False)
    o, g, bad = _den(w["x"].to(DEV), w["lengths"], g2)
    assert bad > 0 and not np.isfinite(o)


def test_numerator_longer_than_its_utterance_is_not_ok():
    """A left-to-right numerator graph with more states than the utterance has frames cannot reach its
    final state: log-likelihood -inf (chain-log-domain-computation.cc:170-190), ok false."""
    D = 40
    graphs = [ChainGraph(syn.make_num_fst(12, D, seed=3), log_domain=True),
              ChainGraph(syn.make_num_fst(4, D, seed=4), log_domain=True)]
    gb = ChainGraphBatch(graphs, max_num_transitions=max(g.num_transitions for g in graphs),
                         max_num_states=max(g.num_states for g in graphs))
    L = torch.tensor([8, 8])                    # 12 states need >= 12 frames; 4 states fit
    x = syn.make_input(2, 8, D, seed=9, device=DEV)
    o, g, bad = _num(x, L, gb)
    assert bad > 0 and o == float("-inf")
    # the sequence that fits still gets its posteriors: rows sum to one
    assert torch.allclose(g[1].sum(-1), torch.ones(8, device=DEV), atol=1e-4)


def test_ok_is_the_third_return_value_of_the_pychain_C_surface():
    w = syn.make_workload("C1")
    from pychain_amd import ChainGraphBatch as GB
    den_b = GB(w["den_graph"], 2)
    x = w["x"].to(DEV)
    bs = torch.nn.utils.rnn.pack_padded_sequence(w["x"], w["lengths"], batch_first=True).batch_sizes
    args = lambda xin: [den_b.forward_transitions.contiguous(), den_b.forward_transition_indices.contiguous(),
                        den_b.forward_transition_probs.contiguous(), den_b.backward_transitions.contiguous(),
                        den_b.backward_transition_indices.contiguous(), den_b.backward_transition_probs.contiguous(),
                        den_b.leaky_probs.contiguous(), den_b.initial_probs.contiguous(), den_b.final_probs.contiguous(),
                        den_b.start_state, xin, bs, w["lengths"], den_b.num_states, 1e-5]
    objf, grad, ok = native.forward_backward(*args(x.clamp(-30, 30).exp()))
    assert ok.dtype == torch.bool and bool(ok.all())
    bad_x = x.clamp(-30, 30).exp()
    bad_x[0, 0, 0] = float("nan")
    objf, grad, ok = native.forward_backward(*args(bad_x))
    assert not bool(ok.all())


# ---- the 5 % invariant itself (chain-computation.cc:363-390, chain-log-domain-computation.cc:289-303) -----------------
# Every test above trips a non-finite branch.  Here the data stay finite and healthy-looking: ONE stored alpha row is scaled
# (option debug_corrupt_row) between the recursions and the occupancy pass, which is exactly what the reference's check is
# there to notice - alpha.beta of that frame no longer equals the sequence's probability.


def _den_with(x, L, den, corrupt=None, verbose=0, **opts):
    ctx = [_lib.option(k, v) for k, v in opts.items()]
    if corrupt:
        ctx.append(_lib.option("debug_corrupt_row", corrupt))
    ctx.append(_lib.option("verbose", verbose))
    for c in ctx:
        c.__enter__()
    try:
        return _den(x, L, den)
    finally:
        for c in reversed(ctx):
            c.__exit__()


@pytest.mark.parametrize("form", ["lazy", "lazy_registers", "two_barrier", "pair", "small"])
def test_five_percent_invariant_fires_denominator(form):
    cfg = syn.CONFIGS["C2" if form == "small" else "C3"]      # (C2's graph runs in four-wave workgroups)
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    L = torch.tensor([90, 77, 64, 90])
    x = syn.make_input(4, 90, cfg["D"], seed=43, device=DEV)
    opts = {"lazy": {}, "lazy_registers": {"den_dma": 0}, "two_barrier": {"den_lazy": 0}, "pair": {"den_pair": 1}, "small": {}}[form]
    o0, g0, bad = _den_with(x, L, den, **opts)
    assert bad == 0
    # frame 0 is checked always (as in the reference)
    o, g, bad = _den_with(x, L, den, corrupt="den,1,0,1.2", **opts)
    assert bad > 0 and o == o0                       # the log-probability comes from the recursions: untouched
    # 4 % is inside the tolerance
    o, g, bad = _den_with(x, L, den, corrupt="den,1,0,1.04", **opts)
    assert bad == 0
    # another frame: seen only when every frame is checked (verbose level >= 1, chain-computation.cc:337-338)
    o, g, bad = _den_with(x, L, den, corrupt="den,2,37,1.2", **opts)
    assert bad == 0
    o, g, bad = _den_with(x, L, den, corrupt="den,2,37,1.2", verbose=1, **opts)
    assert bad > 0
    o, g, bad = _den_with(x, L, den, corrupt="den,2,37,0.8", verbose=1, **opts)
    assert bad > 0
    o, g, bad = _den_with(x, L, den, verbose=1, **opts)
    assert bad == 0 and o == o0 and torch.equal(g, g0)


def test_five_percent_invariant_fires_numerator():
    cfg = syn.CONFIGS["C3"]
    L = torch.tensor([120, 96, 80])
    x = syn.make_input(3, 120, cfg["D"], seed=47, device=DEV)
    numg = syn.make_num_graphs(L.tolist(), cfg["D"], seed=700)

    def run(corrupt=None, verbose=0):
        ctx = [_lib.option("verbose", verbose)] + ([_lib.option("debug_corrupt_row", corrupt)] if corrupt else [])
        for c in ctx:
            c.__enter__()
        try:
            return _num(x, L, numg)
        finally:
            for c in reversed(ctx):
                c.__exit__()
    o0, g0, bad = run()
    assert bad == 0
    o, g, bad = run("num,0,0,1.2")
    assert bad > 0 and o == o0
    o, g, bad = run("num,0,0,1.04")
    assert bad == 0
    o, g, bad = run("num,1,50,1.2")
    assert bad == 0
    o, g, bad = run("num,1,50,1.2", verbose=1)
    assert bad > 0
    # the fused loss reports it in its numerator word
    xx = x.clone().requires_grad_(True)
    den = syn.make_den_graph(200, 2000, cfg["D"], seed=2)
    with _lib.option("debug_corrupt_row", "num,2,0,1.3"):
        ChainLoss(den, 1e-5)(xx, L, numg).backward()
    torch.cuda.synchronize()
    bc = ChainFunction.last_bad_count.tolist()
    assert bc[0] == 0 and bc[1] > 0

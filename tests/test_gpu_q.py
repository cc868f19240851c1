"""One-word state vectors in the lazy recursions (option den_q; den_lazy.inc.h: MAP::kQ; launch-hint bit 19).  The second word of the
float2 {a, coef leaky} / {b, 1} the recursions gather per arc is a constant of the STATE: alpha can gather a / (coef leaky) and keep
p_k coef leaky(src_k) as the arc's constant, beta gathers b alone - two ds_read_b32 per arc instead of a ds_read_b64 and a ds_read_b32
(chain-computation.cc:150-194, :289-330: the same recursions, another association of the same products).  Off by default (measured: it
does not pay); held here to the fp64 oracle and to the default kernels on the same plan."""
import pytest
import torch

import oracle as orc
from helpers import record_parity, rel_err
from pychain_amd import ChainFunction, ChainGraphBatch, ChainLoss, _lib, _plan, synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D = 3456


@pytest.fixture(scope="module")
def den():
    return syn.make_den_graph(3000, 30000, D, seed=0)          # the benchmark graph (BASELINE.json C3)


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
    return float(o.detach()), xx.grad, int(o.bad_count.sum())


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


def test_the_option_selects_the_one_word_form_where_the_plan_allows_it(den):
    plan = _plan.graph_plan(den, D, torch.device(DEV))
    assert (plan.slot_rows >> 19) & 1                                       # every leaky probability positive, one position per state
    assert "one-word states" not in _name(den, 64, fused=True)             # off by default
    assert "one-word states" in _name(den, 64, fused=True, den_q=1) and "one-word states" in _name(den, 8, den_q=1, den_tseg=0)
    assert "one-word states" not in _name(den, 8, den_q=1, den_dma=0)      # rows through registers: the two-word kernels
    sg = syn.make_structured_den_graph()                                    # a pdf-by-state plan keeps its one-gather form
    assert "one-word states" not in _lib.den_kernel_names(_plan.graph_plan(sg, D, torch.device(DEV)).slot_rows, 3000, D, 8)[0]


@pytest.mark.parametrize("lens", [[700, 651, 512, 333, 2, 1], [257, 256, 3], [1]])
@pytest.mark.parametrize("ahead", [0, 1])
def test_one_word_states_vs_oracle_and_vs_the_two_word_kernels(den, lens, ahead):
    """Ragged lengths down to one frame, rows clamped / exp'd by the recursions (den_dma = 2: the form of the fused step) and rows
    exp'd ahead of them; objective and gradient against the fp64 oracle (1e-5) and against the default kernels on the same plan."""
    T, B = max(lens), len(lens)
    L = torch.tensor(lens)
    x = syn.make_input(B, T, D, seed=190 + T, device=DEV)
    dma = {"den_dma": 3} if ahead else {"den_dma": 2}
    o, g, bad = _call(den, x, L, den_q=1, den_tseg=0, **dma)
    o0, g0, bad0 = _call(den, x, L, den_q=0, den_tseg=0, **dma)
    ro, rg = orc.chain_function(x.cpu(), L, ChainGraphBatch(den, B), 1e-5, flavour="f64")
    e, e0 = rel_err(g.cpu().numpy(), rg), rel_err(g.cpu().numpy(), g0.cpu().numpy())
    record_parity("one_word_states_T%d_B%d_ahead%d" % (T, B, ahead), objf=abs(o - ro) / abs(ro), grad_vs_f64=e, grad_vs_two_word_kernels=e0, bound=1e-5)
    assert bad == 0 and bad0 == 0
    assert abs(o - ro) <= 1e-5 * abs(ro) and e <= 1e-5, (o, ro, e)
    assert abs(o - o0) <= 1e-5 * abs(o0) and e0 <= 1e-5, (o, o0, e0)
    o2, g2, _ = _call(den, x, L, den_q=1, den_tseg=0, **dma)               # deterministic: the same call gives the same bits
    assert o2 == o and torch.equal(g2, g)


def test_one_word_states_in_the_fused_loss_and_with_a_nan(den):
    L = torch.tensor([640, 600, 333, 64])
    x = syn.make_input(4, 640, D, seed=23, device=DEV)
    num = syn.make_num_graphs(L.tolist(), D, seed=100)
    with _lib.option("den_q", 1), _lib.option("den_tseg", 0):
        xx = x.clone().requires_grad_(True)
        loss = ChainLoss(den, 1e-5, avg=True)(xx, L, num)
        loss.backward()
        torch.cuda.synchronize()
        assert int(loss.bad_count.sum()) == 0
        rl, rg = orc.chain_loss(x.cpu(), L, den, num, 1e-5, avg=True, flavour="f64")
        assert abs(float(loss.detach()) - rl) <= 1e-5 * abs(rl) and rel_err(xx.grad.cpu().numpy(), rg) <= 1e-5
        xn = x.clone()
        xn[1, 17, 5] = float("nan")
        o, g, bad = _call(den, xn, L, den_q=1, den_tseg=0)
        assert bad > 0 and o != o                                           # `ok` false, NaN objective: as the two-word kernels report it

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

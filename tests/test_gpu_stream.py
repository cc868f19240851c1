"""The streamed occupancy pass (DenArgs::stream; den_kernels.hip: stream_take): one persistent launch that draws rings
of frames from a queue in the order in which the two recursions of a sequence make them computable, instead of gated time
segments.  Same frames, same pairing: bit-identical to the segmented schedule - for ragged lengths down to one frame, odd
and even middles, the one-frame kernel (rows wider than 4096 pdfs), the fold of the numerator, two sequences per
recursion workgroup; and against the oracle.  Replaces the sequencing of chain-computation.cc:345-391 (Backward)."""
import numpy as np
import pytest
import torch

import oracle as orc
from helpers import rel_err
from pychain_amd import ChainFunction, ChainGraphBatch, ChainLoss, _lib, _plan, synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _den(x, L, den, **opts):
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
    return o.detach().clone(), xx.grad, int(ChainFunction.last_bad_count.sum())


@pytest.mark.parametrize("H,K,D", [(3000, 30000, 3456), (300, 2500, 512), (700, 6000, 8408), (200, 2000, 1000)])
def test_streamed_equals_segmented_and_the_oracle(H, K, D):
    den = syn.make_den_graph(H, K, D, seed=3)
    lengths = [411, 410, 409, 333, 256, 255, 97, 33, 32, 31, 17, 16, 15, 2, 1]
    L = torch.tensor(lengths)
    x = syn.make_input(len(lengths), 411, D, seed=61, device=DEV)
    x[3, 333:] = float("nan")                                   # padding frames may hold anything
    o0, g0, bad0 = _den(x, L, den, den_segments=1)
    o1, g1, bad1 = _den(x, L, den, den_segments=3)
    for rep in range(3):                                        # (the queue is drawn in a different interleaving every time)
        o2, g2, bad2 = _den(x, L, den)
        assert bad0 == 0 and bad1 == 0 and bad2 == 0
        assert torch.equal(o0, o1) and torch.equal(g0, g1) and torch.equal(o0, o2) and torch.equal(g0, g2), rep
    live = (torch.arange(411)[None, :] < L[:, None]).to(DEV)
    assert bool((g2[~live] == 0).all())
    ro, rg = orc.chain_function(torch.nan_to_num(x.cpu(), nan=0.0), L, ChainGraphBatch(den, len(lengths)), 1e-5)
    assert abs(float(o2) - ro) <= 1e-4 * abs(ro) and rel_err(g2.cpu().numpy(), rg) <= 1e-4


def test_streamed_fused_loss_with_the_numerator_folded_in():
    cfg = syn.CONFIGS["C3"]
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    L = torch.tensor([520, 519, 300, 300, 299, 64, 9])
    x = syn.make_input(7, 520, cfg["D"], seed=67, device=DEV)
    numg = syn.make_num_graphs(L.tolist(), cfg["D"], seed=800)

    def run(**opts):
        ctx = [_lib.option(k, v) for k, v in opts.items()]
        for c in ctx:
            c.__enter__()
        try:
            xx = x.clone().requires_grad_(True)
            loss = ChainLoss(den, 1e-5, avg=False)(xx, L, numg)
            loss.backward()
            torch.cuda.synchronize()
            return loss.detach().clone(), xx.grad, ChainFunction.last_bad_count.tolist()
        finally:
            for c in reversed(ctx):
                c.__exit__()
    l0, g0, b0 = run(den_segments=3)
    for rep in range(3):
        l1, g1, b1 = run()
        assert b0 == [0, 0] and b1 == [0, 0] and torch.equal(l0, l1) and torch.equal(g0, g1), rep
    # the one-frame occupancy kernel, numerator accumulated into the stored gradient afterwards: the same values to rounding
    # (16 waves instead of 8 sum a frame's total in another order)
    l2, g2, b2 = run(gamma16=1)
    assert torch.equal(l0, l2) and rel_err(g2.cpu().numpy(), g0.cpu().numpy()) <= 1e-6
    rl, rg = orc.chain_loss(x.cpu(), L, den, numg, 1e-5, avg=False, flavour="f64")
    assert abs(float(l1) - float(rl)) <= 1e-4 * abs(float(rl)) and rel_err(g1.cpu().numpy(), rg) <= 1e-5


def test_streamed_with_the_check_on_every_frame_and_a_broken_invariant():
    """verbose >= 1 checks every frame after both passes are done (no overlap, no stream); at level 0 the streamed pass
    records frame 0 and the corrupted-row hook still fires."""
    den = syn.make_den_graph(300, 2500, 512, seed=5)
    L = torch.tensor([300, 290, 280])
    x = syn.make_input(3, 300, 512, seed=71, device=DEV)
    o0, g0, bad0 = _den(x, L, den)
    o1, g1, bad1 = _den(x, L, den, verbose=1)
    assert bad0 == 0 and bad1 == 0 and torch.equal(o0, o1) and torch.equal(g0, g1)
    o2, g2, bad2 = _den(x, L, den, debug_corrupt_row="den,1,0,1.3")
    assert bad2 > 0

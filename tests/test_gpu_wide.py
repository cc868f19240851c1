"""The 8-wave shape of the lazy-normalisation recursion (den_lazy.inc.h: LzWide; nnet-output rows of up to 9216 pdfs,
the C4 path) and the choice of kernel per shape.  Replaces chain-computation.cc:113-194,247-330 at D > 4096."""
import numpy as np
import pytest
import torch

import oracle as orc
from helpers import rel_err
from pychain_amd import ChainFunction, ChainGraphBatch, _lib, _plan, synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _den(x, L, den, **opts):
    xx = x.clone().requires_grad_(True)
    gb = ChainGraphBatch(den, x.size(0))
    ctx = [_lib.option(k, v) for k, v in opts.items()]
    for c in ctx:
        c.__enter__()
    try:
        o = ChainFunction.apply(xx, L, gb, 1e-5)
        o.backward()
        torch.cuda.synchronize()
    finally:
        for c in reversed(ctx):
            c.__exit__()
    assert int(ChainFunction.last_bad_count.sum()) == 0
    return float(o.detach()), xx.grad


def _names(den, D, B, **opts):
    plan = _plan.graph_plan(den, D, torch.device(DEV))
    ctx = [_lib.option(k, v) for k, v in opts.items()]
    for c in ctx:
        c.__enter__()
    try:
        return _lib.den_kernel_names(plan.slot_rows, den.num_states, D, B)
    finally:
        for c in reversed(ctx):
            c.__exit__()


def test_wide_recursion_on_the_c3_graph_vs_its_sixteen_wave_form_and_the_oracle():
    """Forced onto the C3 graph (where the 16-wave shape is the default): same results to rounding as the 16-wave form,
    bit-identical to itself over every occupancy schedule, and within 1e-4 of the oracle; ragged lengths, a one-frame
    sequence."""
    cfg = syn.CONFIGS["C3"]
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    L = torch.tensor([301, 288, 130, 1])
    x = syn.make_input(4, 301, cfg["D"], seed=21, device=DEV)
    assert _names(den, cfg["D"], 4, den_dma=0)[0] == "den_recursion_lazy_kernel"
    assert _names(den, cfg["D"], 4, den_wide=1)[0] == "den_recursion_lazy_kernel<wide>"
    o16, g16 = _den(x, L, den, den_dma=0)
    o8, g8 = _den(x, L, den, den_wide=1)
    assert abs(o8 - o16) <= 1e-6 * abs(o16) and rel_err(g8.cpu().numpy(), g16.cpu().numpy()) <= 1e-5
    for nseg in (1, 3):
        o, g = _den(x, L, den, den_wide=1, den_segments=nseg)
        assert o == o8 and torch.equal(g, g8)
    ro, rg = orc.chain_function(x.cpu(), L, ChainGraphBatch(den, 4), 1e-5)
    assert abs(o8 - ro) <= 1e-4 * abs(ro) and rel_err(g8.cpu().numpy(), rg) <= 1e-4
    assert bool((g8[2, 130:] == 0).all()) and bool((g8[3, 1:] == 0).all())


def test_two_copy_recursion_on_the_c3_graph_vs_the_one_copy_form_and_the_oracle(monkeypatch):
    """An experiment kept reproducible (measured 1.2 % slower, profiles/r03_i_two_copies.txt): under PYCHAIN_PLAN_CHOICE=1 C3's
    plan holds two-copy tiles, and the recursion then keeps the nnet-output row twice in LDS, every arc reading the copy the plan
    picked.  Same results to rounding as the one-copy kernel (option den_two_copy = 0), bit-identical over the occupancy
    schedules, within 1e-4 of the oracle; ragged lengths, a one-frame sequence, a NaN."""
    monkeypatch.setenv("PYCHAIN_PLAN_CHOICE", "1")
    monkeypatch.setenv("PYCHAIN_PLAN_CACHE_DIR", "off")
    cfg = syn.CONFIGS["C3"]
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    L = torch.tensor([301, 288, 130, 1])
    x = syn.make_input(4, 301, cfg["D"], seed=21, device=DEV)
    assert _names(den, cfg["D"], 4)[0] == "den_recursion_lazy_kernel<two copies>"
    assert _names(den, cfg["D"], 4, den_two_copy=0)[0] == "den_recursion_lazy_kernel<dma>"
    o2, g2 = _den(x, L, den)
    o1, g1 = _den(x, L, den, den_two_copy=0)
    assert abs(o2 - o1) <= 1e-6 * abs(o1) and rel_err(g2.cpu().numpy(), g1.cpu().numpy()) <= 1e-5
    for nseg in (1, 3):
        o, g = _den(x, L, den, den_stream=0, den_segments=nseg)
        assert o == o2 and torch.equal(g, g2)
    ro, rg = orc.chain_function(x.cpu(), L, ChainGraphBatch(den, 4), 1e-5)
    assert abs(o2 - ro) <= 1e-4 * abs(ro) and rel_err(g2.cpu().numpy(), rg) <= 1e-4
    assert bool((g2[2, 130:] == 0).all()) and bool((g2[3, 1:] == 0).all())
    x[1, 17, cfg["D"] - 1] = float("nan")
    xx = x.clone().requires_grad_(True)
    o = ChainFunction.apply(xx, L, ChainGraphBatch(den, 4), 1e-5)
    torch.cuda.synchronize()
    assert int(ChainFunction.last_bad_count.sum()) > 0 and np.isnan(float(o.detach()))


def test_twelve_wave_experiment_on_the_c3_graph(monkeypatch):
    """den_wide = 2: the 12-wave dealing (in the plan only under PYCHAIN_PLAN_TWELVE=1), 56-row loops, 163 VGPRs.  Measured
    15 % slower than the 16-wave kernel (profiles/r03_g_twelve_waves.txt) - kept as a reproducible experiment: same
    results to rounding, within 1e-4 of the oracle; on a plan without the dealing the kernel refuses (ok = False)."""
    cfg = syn.CONFIGS["C3"]
    L = torch.tensor([301, 288, 130, 1])
    x = syn.make_input(4, 301, cfg["D"], seed=21, device=DEV)
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    o16, g16 = _den(x, L, den)
    xx = x.clone().requires_grad_(True)
    with _lib.option("den_wide", 2):
        ChainFunction.apply(xx, L, ChainGraphBatch(den, 4), 1e-5)
    torch.cuda.synchronize()
    assert int(ChainFunction.last_bad_count.sum()) > 0
    monkeypatch.setenv("PYCHAIN_PLAN_TWELVE", "1")
    monkeypatch.setenv("PYCHAIN_PLAN_CACHE_DIR", "off")
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    assert _names(den, cfg["D"], 4, den_wide=2)[0] == "den_recursion_lazy_kernel<12 waves>"
    o12, g12 = _den(x, L, den, den_wide=2)
    assert abs(o12 - o16) <= 1e-6 * abs(o16) and rel_err(g12.cpu().numpy(), g16.cpu().numpy()) <= 1e-5
    ro, rg = orc.chain_function(x.cpu(), L, ChainGraphBatch(den, 4), 1e-5)
    assert abs(o12 - ro) <= 1e-4 * abs(ro) and rel_err(g12.cpu().numpy(), rg) <= 1e-4


@pytest.mark.parametrize("H,K,D", [(40, 300, 4100), (700, 6000, 8408), (3000, 30000, 9216)])
def test_wide_rows_vs_oracle(H, K, D):
    """4096 < D <= 9216 through the 8-wave recursion (small, medium and C4-size graphs: 32-, 64- and 80-row loops):
    against the oracle."""
    den = syn.make_den_graph(H, K, D, seed=3)
    L = torch.tensor([97, 64, 5])
    x = syn.make_input(3, 97, D, seed=33, device=DEV)
    assert _names(den, D, 3, den_wide=1)[0] == "den_recursion_lazy_kernel<wide>"
    o, g = _den(x, L, den, den_wide=1)
    ro, rg = orc.chain_function(x.cpu(), L, ChainGraphBatch(den, 3), 1e-5)
    assert abs(o - ro) <= 1e-4 * abs(ro) and rel_err(g.cpu().numpy(), rg) <= 1e-4
    o2, g2 = _den(x, L, den, den_wide=0, den_dma=0)          # the two-barrier kernel: second opinion
    assert abs(o - o2) <= 1e-6 * abs(o2) and rel_err(g.cpu().numpy(), g2.cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("H,K,D", [(40, 300, 4100), (700, 6000, 8408), (3000, 30000, 9216), (3000, 30000, 8408),
                                   (40, 300, 3457), (200, 2000, 1001), (300, 3000, 4098), (64, 500, 7)])
def test_dma_rows_vs_oracle(H, K, D):
    """The 16-wave lazy recursion with LDS-direct nnet-output rows is the default wherever a lazy shape fits: rows of
    4096 < D <= 9216 pdfs (small, medium and C4-size graphs: 16-, 32- and 40-row loops), a row whose last 1 KiB chunk is
    partial, and row lengths that are NOT a multiple of four floats (rows then start at any 4-byte address: the last lane
    of a row reads a few floats of the next one, which must not leak - not even as a false NaN alarm); ragged lengths;
    against the oracle and against the two-barrier kernel."""
    den = syn.make_den_graph(H, K, D, seed=3)
    L = torch.tensor([97, 64, 5, 1])
    x = syn.make_input(4, 97, D, seed=33, device=DEV)
    x[1, 64:] = float("nan")                                 # padding frames may hold anything
    x[2, 5:] = float("nan")
    assert _names(den, D, 4)[0] == "den_recursion_lazy_kernel<dma>"
    o, g = _den(x, L, den)
    ro, rg = orc.chain_function(torch.nan_to_num(x.cpu(), nan=0.0), L, ChainGraphBatch(den, 4), 1e-5)
    assert abs(o - ro) <= 1e-4 * abs(ro) and rel_err(g.cpu().numpy(), rg) <= 1e-4
    o2, g2 = _den(x, L, den, den_dma=0, den_lazy=0)          # the two-barrier kernel: second opinion
    assert _names(den, D, 4, den_dma=0, den_lazy=0)[0] == "den_recursion_kernel"
    assert abs(o - o2) <= 1e-6 * abs(o2) and rel_err(g.cpu().numpy(), g2.cpu().numpy()) <= 1e-5
    for nseg in (1, 3):
        o3, g3 = _den(x, L, den, den_segments=nseg)
        assert o3 == o and torch.equal(g3, g)


def test_dma_rows_on_the_narrow_map_match_the_register_path_bit_for_bit():
    """The same recursion with its rows through LDS-direct loads (default) and through registers (option den_dma = 0):
    the arithmetic is the same operation for operation."""
    cfg = syn.CONFIGS["C3"]
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    L = torch.tensor([301, 288, 130, 1])
    x = syn.make_input(4, 301, cfg["D"], seed=21, device=DEV)
    o, g = _den(x, L, den, den_dma=0)
    o2, g2 = _den(x, L, den)
    assert _names(den, cfg["D"], 4)[0] == "den_recursion_lazy_kernel<dma>"
    assert o == o2 and torch.equal(g, g2)
    xn = x.clone()
    xn[1, 200, 77] = float("nan")
    xx = xn.requires_grad_(True)
    o = ChainFunction.apply(xx, L, ChainGraphBatch(den, 4), 1e-5)
    torch.cuda.synchronize()
    assert int(ChainFunction.last_bad_count.sum()) > 0 and np.isnan(float(o.detach()))


def test_wide_recursion_nan_and_bad_lengths():
    den = syn.make_den_graph(60, 400, 5000, seed=5)
    x = syn.make_input(2, 40, 5000, seed=7, device=DEV)
    x[1, 17, 4999] = float("nan")
    xx = x.clone().requires_grad_(True)
    with _lib.option("den_wide", 1):
        o = ChainFunction.apply(xx, torch.tensor([40, 33]), ChainGraphBatch(den, 2), 1e-5)
    torch.cuda.synchronize()
    assert int(ChainFunction.last_bad_count.sum()) > 0 and np.isnan(float(o.detach()))


def test_which_kernel_each_shape_gets():
    """The kernel a shape selects is pinned (a silent drop to a slower kernel is a performance bug nobody sees):
    pychain_hip_den_kernel_names answers from the same predicates the launcher uses."""
    cases = [
        # H, K, D, B -> recursion, occupancy
        (3000, 30000, 3456, 64, "den_recursion_lazy_kernel<dma>", "den_gamma2_kernel"),     # C3
        (3000, 30000, 3456, 128, "den_recursion_pair_kernel", "den_gamma2_kernel"),         # B >= 96: two sequences per workgroup
        (200, 2000, 1000, 64, "den_recursion_lazy_kernel<dma>", "den_gamma2_kernel"),       # C2
        (20, 60, 40, 2, "den_recursion_lazy_kernel<dma>", "den_gamma2_kernel"),             # C1
        (3000, 30000, 8408, 32, "den_recursion_lazy_kernel<dma>", "den_gamma_kernel"),      # C4
        (300, 3000, 4100, 8, "den_recursion_lazy_kernel<dma>", "den_gamma_kernel"),
        (300, 3000, 9220, 8, "den_recursion_kernel", "den_gamma_kernel"),                   # rows beyond the LDS map
        (300, 3000, 4098, 8, "den_recursion_lazy_kernel<dma>", "den_gamma_kernel"),         # D % 4 != 0: rows by LDS-direct loads
        (3000, 30000, 3457, 64, "den_recursion_lazy_kernel<dma>", "den_gamma_kernel"),      # one pdf more than C3: no cliff in the recursion
        (5000, 20000, 3456, 8, "den_recursion_kernel", "den_gamma_kernel"),                 # more states than the lazy LDS maps hold
    ]
    for H, K, D, B, rec, occ in cases:
        den = syn.make_den_graph(H, K, D, seed=1)
        got = _names(den, D, B)
        assert got[0] == rec, (H, K, D, B, got)
        if occ is not None:
            assert got[1] == occ, (H, K, D, B, got)

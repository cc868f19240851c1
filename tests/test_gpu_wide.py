"""The shapes of the lazy-normalisation recursion (den_lazy.inc.h: 16 waves with rows through registers or by LDS-direct
loads up to 9216 pdfs - C3, C4 - and FOUR waves for small graphs - C1, C2) and the choice of kernel per shape.
Replaces chain-computation.cc:113-194,247-330."""
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


def _names(den, D, B, fused=False, **opts):
    plan = _plan.graph_plan(den, D, torch.device(DEV))
    ctx = [_lib.option(k, v) for k, v in opts.items()]
    for c in ctx:
        c.__enter__()
    try:
        return _lib.den_kernel_names(plan.slot_rows, plan.num_states, D, B, fused=fused)
    finally:
        for c in reversed(ctx):
            c.__exit__()


@pytest.mark.parametrize("H,K,D,T", [(200, 2000, 1000, 150), (20, 60, 40, 50), (600, 5000, 2048, 97), (1000, 7000, 4096, 64), (130, 900, 1001, 33)])
def test_small_graphs_run_in_four_wave_workgroups(H, K, D, T):
    """Graphs whose recursion tiles fit four waves (C2, C1, up to 1024 states / 40 rows per wave; any row length) run
    den_recursion_lazy_kernel<small> over the plan's four-wave dealing: against the oracle, against the 16-wave lazy form
    (option den_dma = 0: same rows and slot order, another dealing - equal to rounding) and the two-barrier kernel, bit-identical
    to itself over the occupancy schedules; ragged lengths, a one-frame sequence, a NaN."""
    den = syn.make_den_graph(H, K, D, seed=3)
    L = torch.tensor([T, max(1, (2 * T) // 3), 5, 1])
    x = syn.make_input(4, T, D, seed=33, device=DEV)
    x[1, L[1]:] = float("nan")                               # padding frames may hold anything
    assert _names(den, D, 4)[0] == "den_recursion_lazy_kernel<small>"
    assert _names(den, D, 256)[0] == "den_recursion_lazy_kernel<small>"      # several workgroups to a CU: no pair kernel
    assert _names(den, D, 4, den_dma=0)[0] == "den_recursion_lazy_kernel" if D % 4 == 0 else True
    o, g = _den(x, L, den)
    ro, rg = orc.chain_function(torch.nan_to_num(x.cpu(), nan=0.0), L, ChainGraphBatch(den, 4), 1e-5)
    assert abs(o - ro) <= 1e-4 * abs(ro) and rel_err(g.cpu().numpy(), rg) <= 1e-4
    for opts in ({"den_dma": 0}, {"den_lazy": 0}):
        o2, g2 = _den(x, L, den, **opts)
        assert abs(o - o2) <= 1e-6 * abs(o2) and rel_err(g.cpu().numpy(), g2.cpu().numpy()) <= 1e-5
    for nseg in (1, 3):
        o3, g3 = _den(x, L, den, den_segments=nseg)
        assert o3 == o and torch.equal(g3, g)
    assert bool((g[2, 5:] == 0).all()) and bool((g[3, 1:] == 0).all())
    xn = torch.nan_to_num(x, nan=0.0)
    xn[0, T // 2, D - 1] = float("nan")
    xx = xn.requires_grad_(True)
    ob = ChainFunction.apply(xx, L, ChainGraphBatch(den, 4), 1e-5)
    torch.cuda.synchronize()
    assert int(ChainFunction.last_bad_count.sum()) > 0 and np.isnan(float(ob.detach()))


@pytest.mark.parametrize("H,K,D", [(40, 300, 4100), (700, 6000, 8408), (3000, 30000, 9216), (3000, 30000, 8408),
                                   (1400, 9000, 3457), (1100, 12000, 1001), (300, 3000, 4098), (1200, 5000, 7)])
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


@pytest.mark.parametrize("name,H,K,D,T,lens", [
    ("C3", 3000, 30000, 3456, 301, [301, 288, 130, 1, 2, 65]),       # 16 waves, rows of <= 4096 pdfs
    ("C4", 3000, 30000, 8408, 160, [160, 159, 64, 3]),               # rows of up to 9216 pdfs
    ("C2", 200, 2000, 1000, 150, [150, 149, 75, 1, 64, 150]),        # four-wave workgroups
])
def test_rows_exp_ahead_of_the_recursions_are_the_same_rows(name, H, K, D, T, lens):
    """den_exp_rows_kernel (DenArgs::ex) writes exp(clamp(x)) from both ends of every sequence towards its middle while
    the recursions run, and they gather from those rows as they arrive instead of passing over each row in LDS (option
    den_dma = 2: the old way).  Same clamp_exp, same recursion: objf and gradient bit for bit, for every length (one
    frame, even / odd middles), streamed, gated and without overlap; a NaN network output is still seen
    (loss.py:30,43: it reaches the loss) wherever it sits."""
    den = syn.make_den_graph(H, K, D, seed=0)
    L = torch.tensor(lens)
    x = syn.make_input(len(lens), T, D, seed=29, device=DEV)
    for extra in ({}, {"den_segments": 3}, {"den_segments": 1}):
        o, g = _den(x, L, den, den_dma=2, **extra)
        o2, g2 = _den(x, L, den, den_dma=3, **extra)       # (3: also where the default no longer exp's ahead - rows of <= 4096 pdfs)
        assert o == o2 and torch.equal(g, g2), extra
    for (b, t, d) in ((0, 0, 5), (0, lens[0] // 2, D - 1), (1, lens[1] - 1, 0), (2, 7, 33)):
        xn = x.clone()
        xn[b, t, d] = float("nan")
        xx = xn.requires_grad_(True)
        o = ChainFunction.apply(xx, L, ChainGraphBatch(den, len(lens)), 1e-5)
        torch.cuda.synchronize()
        assert int(ChainFunction.last_bad_count.sum()) > 0 and np.isnan(float(o.detach())), (b, t, d)
    # a NaN past the end of its sequence is nobody's business
    xn = x.clone()
    xn[2, lens[2], 3] = float("nan")
    o3, g3 = _den(xn, L, den)
    assert o3 == o2 and torch.equal(g3, g2)


def test_which_kernel_each_shape_gets():
    """The kernel a shape selects is pinned (a silent drop to a slower kernel is a performance bug nobody sees):
    pychain_hip_den_kernel_names answers from the same predicates the launcher uses."""
    cases = [
        # H, K, D, B -> recursion, occupancy
        (3000, 30000, 3456, 64, "den_recursion_lazy_kernel<dma>", "den_gamma2_kernel"),     # C3
        (3000, 30000, 3456, 128, "den_recursion_lazy_kernel<dma>", "den_gamma2_kernel"),    # the denominator alone: one-sequence workgroups while 2 B fit the chip
        (3000, 30000, 3456, 160, "den_recursion_pair_kernel", "den_gamma2_kernel"),         # ... then two sequences per workgroup
        (200, 2000, 1000, 64, "den_recursion_lazy_kernel<small>", "den_gamma2_kernel"),     # C2: four-wave workgroups
        (20, 60, 40, 2, "den_recursion_lazy_kernel<small>", "den_gamma2_kernel"),           # C1
        (200, 2000, 1000, 256, "den_recursion_lazy_kernel<small>", "den_gamma2_kernel"),    # ... also at batch sizes that pair large graphs
        (900, 6000, 4096, 8, "den_recursion_lazy_kernel<small>", "den_gamma2_kernel"),      # about the largest graphs four waves take
        (1100, 8000, 4096, 8, "den_recursion_lazy_kernel<dma>", "den_gamma2_kernel"),       # more than 1024 states: 16 waves
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
    # a fused loss leaves part of the chip to its numerator: pairs from B = 100 on (256 CUs)
    den = syn.make_den_graph(3000, 30000, 3456, seed=1)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert _names(den, 3456, (25 * cus + 63) // 64, fused=True)[0] == "den_recursion_pair_kernel"
    assert _names(den, 3456, (25 * cus + 63) // 64 - 4, fused=True)[0] == "den_recursion_lazy_kernel<dma>"


def test_structured_phone_lm_like_graph_vs_oracle():
    """synthetic.make_structured_den_graph (arcs entering a state share its pdf, strong self-loops - what a phone LM composed
    with the chain topology looks like; the benchmark graph is random): the C3-size one through the default kernels against
    the oracle, and a small one through the four-wave shape."""
    for n, fan, D, T, lens in ((1500, 9, 3456, 200, [200, 150, 64, 1]), (90, 5, 1000, 80, [80, 33, 2])):
        den = syn.make_structured_den_graph(n, fan, D)
        L = torch.tensor(lens)
        x = syn.make_input(len(lens), T, D, seed=61, device=DEV)
        o, g = _den(x, L, den)
        ro, rg = orc.chain_function(x.cpu(), L, ChainGraphBatch(den, len(lens)), 1e-5)
        assert abs(o - ro) <= 1e-4 * abs(ro) and rel_err(g.cpu().numpy(), rg) <= 1e-4, (n, rel_err(g.cpu().numpy(), rg))


@pytest.mark.parametrize("case", ["small_hubs", "hubs", "structured"])
def test_states_on_several_lanes_on_the_device(case):
    """Plans that put states with many arcs on several positions (csrc/plan.cpp; tests/test_plan.py holds the CPU side):
    every recursion family that reads such a plan - lazy with LDS-direct rows, lazy with rows through registers, the
    two-barrier kernel, two sequences per workgroup, time segments - against the fp64 oracle and against the same graph
    compiled with every state on one lane."""
    from test_plan import _hub_graph
    if case == "small_hubs":
        (den, D), T = _hub_graph(seed=6, H=200, D=300, extra=1500, hubs=3, fan=45), 120
    elif case == "hubs":
        (den, D), T = _hub_graph(), 420
    else:
        den, D, T = syn.make_structured_den_graph(), 3456, 420
    dev = torch.device(DEV)
    plan = _plan.graph_plan(den, D, dev)
    with _lib.option("plan_split", "0"):
        plan0 = _plan.graph_plan(den, D, dev)
    assert plan.num_states > den.num_states == plan0.num_states and (plan.slot_rows & 1023) < (plan0.slot_rows & 1023)
    L = torch.tensor([T, T - 7, (2 * T) // 3, 3, 1])
    x = syn.make_input(5, T, D, seed=71, device=DEV)
    ro, rg = orc.chain_function(x.cpu(), L, ChainGraphBatch(den, 5), 1e-5, flavour="f64")
    with _lib.option("plan_split", "0"):
        o0, g0 = _den(x, L, den)
    forms = [{}, {"den_dma": 0}, {"den_lazy": 0}, {"den_pair": 1}, {"den_pair": 1, "den_lazy": 0}]
    if case != "small_hubs":
        forms.append({"den_tseg": 2, "den_tburn": 96})          # (T = 420: two segments of 210 frames that start 96 outside)
    from helpers import record_parity
    worst = {}
    for opts in forms:
        o, g = _den(x, L, den, **opts)
        e = rel_err(g.cpu().numpy(), rg)
        assert abs(o - ro) <= 1e-5 * abs(ro) and e <= 2e-5, (case, opts, o, ro, e)
        e0 = rel_err(g.cpu().numpy(), g0.cpu().numpy())
        assert abs(o - o0) <= 1e-5 * abs(o0) and e0 <= 2e-5, (case, opts)
        worst["grad_vs_f64"] = max(worst.get("grad_vs_f64", 0.0), e)
        worst["grad_vs_every_state_on_one_lane"] = max(worst.get("grad_vs_every_state_on_one_lane", 0.0), e0)
    record_parity("states_on_several_lanes_" + case, positions=plan.num_states, states=den.num_states, **worst)

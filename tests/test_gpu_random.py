"""Seeded random shapes through all three denominator recursion kernels (lazy, one-sequence two-barrier, two
sequences per workgroup) against the oracle and against each other: state counts that do not fill their last
group, arc counts from sparse to dense (16 / 32 / 40 resident slot-rows), pdf counts that are not a multiple of
the workgroup's float4 width, batches of every parity, lengths from one frame up."""
import numpy as np
import pytest
import torch

import oracle as orc
from helpers import rel_err
from pychain_amd import ChainFunction, ChainGraphBatch, _lib, synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4          # BASELINE.json north_star: objf and gradient within 1e-4 relative of the fp32 reference arithmetic


def _run(x, L, den, lazy, pair):
    lib = _lib.lib()
    lib.pychain_hip_set_den_lazy(lazy)
    try:
        with _lib.option("den_pair", "1" if pair else "0"):
            xx = torch.as_tensor(x).to(DEV).requires_grad_(True)
            objf = ChainFunction.apply(xx, L, ChainGraphBatch(den, x.shape[0]), 1e-5)
            objf.backward()
            torch.cuda.synchronize()
            assert int(ChainFunction.last_bad_count.sum().item()) == 0
            return float(objf.detach()), xx.grad.cpu().numpy()
    finally:
        lib.pychain_hip_set_den_lazy(1)


@pytest.mark.parametrize("seed", range(12))
def test_random_shapes_all_recursion_kernels(seed):
    rng = np.random.RandomState(1000 + seed)
    H = int(rng.choice([17, 64, 100, 257, 640, 1000]))
    K = int(H * rng.choice([2, 5, 12, 25]))                    # arcs per state: sparse ... dense
    D = int(rng.choice([40, 44, 132, 500, 1028]))
    B = int(rng.randint(1, 8))
    T = int(rng.randint(3, 41))
    lengths = sorted((int(rng.randint(1, T + 1)) for _ in range(B)), reverse=True)
    lengths[0] = T
    den = syn.make_den_graph(H, K, D, seed=seed)
    x = syn.make_input(B, T, D, seed=50 + seed)
    L = torch.tensor(lengths)
    ro, rg = orc.chain_function(x, L, ChainGraphBatch(den, B), 1e-5)
    results = {}
    for name, lazy, pair in (("lazy", 1, False), ("two-barrier", 0, False), ("pair", 0, True)):
        o, g = _run(x, L, den, lazy, pair)
        assert abs(o - ro) <= TOL * abs(ro), (name, H, K, D, B, lengths, o, ro)
        assert rel_err(g, rg) <= TOL, (name, H, K, D, B, lengths, rel_err(g, rg))
        for b, l in enumerate(lengths):
            assert np.all(g[b, l:] == 0.0), (name, b)              # padding frames: exact zeros
        results[name] = (o, g)
    if B >= 2:                                                     # (B = 1 never pairs: the same kernel ran twice)
        assert results["pair"][0] == results["two-barrier"][0] and np.array_equal(results["pair"][1], results["two-barrier"][1])


@pytest.mark.parametrize("seed", range(10))
def test_random_shapes_rows_exp_ahead(seed):
    """den_exp_rows_kernel (rows exp'd ahead of the lazy recursions from both ends of every sequence) against
    the recursions exp'ing their rows themselves (option den_dma = 2), bit for bit: random batch sizes (1 .. 2 workgroups per
    end up to 4), lengths from one frame to T with T from 64 up, pdf counts of every chunk count of the launch, repeated calls
    on one workspace (the counters of the previous call must be gone)."""
    rng = np.random.RandomState(7000 + seed)
    H, K = [(200, 2000), (640, 6000), (3000, 30000)][seed % 3]
    D = int(rng.choice([512, 1000, 2048, 3456, 4100, 6144, 8408]))
    if (H, D) == (200, 8408):
        D = 1000
    B = int(rng.choice([1, 2, 3, 5, 8, 33, 64]))
    T = int(rng.choice([64, 65, 97, 130, 257]))
    if B * T * D > 40e6:
        B = max(1, int(40e6 // (T * D)))
    lengths = [int(rng.randint(1, T + 1)) for _ in range(B)]
    lengths[0] = T
    den = syn.make_den_graph(H, K, D, seed=seed)
    x = syn.make_input(B, T, D, seed=90 + seed, device=DEV)
    L = torch.tensor(lengths)

    def run(**opts):
        xx = x.clone().requires_grad_(True)
        ctx = [_lib.option(k, v) for k, v in opts.items()]
        for c in ctx:
            c.__enter__()
        try:
            o = ChainFunction.apply(xx, L, ChainGraphBatch(den, B), 1e-5)
            o.backward()
            torch.cuda.synchronize()
        finally:
            for c in reversed(ctx):
                c.__exit__()
        assert int(ChainFunction.last_bad_count.sum()) == 0
        return float(o.detach()), xx.grad
    o0, g0 = run(den_dma=2)
    for _ in range(3):
        o, g = run(den_dma=3)
        assert o == o0 and torch.equal(g, g0), (H, K, D, B, T, lengths[:8])


@pytest.mark.parametrize("seed", range(10))
def test_random_shapes_two_byte_rows(seed):
    """bf16 / fp16 network outputs read by the kernels against the host-side up-cast, BIT FOR BIT, on random
    shapes: every recursion family that takes 2-byte rows (lazy with LDS-direct rows in its three maps, four-wave, pair), both
    occupancy kernels, rows exp'd ahead or not, fused loss and numerator alone; pdf counts of every chunk count (multiples of 8),
    batches of every parity, lengths from one frame up; shapes the library declines are up-cast and still agree."""
    from pychain_amd import ChainLoss, _plan, native
    rng = np.random.RandomState(9000 + seed)
    H, K = [(100, 700), (200, 2000), (640, 6000), (1100, 9000), (3000, 30000)][seed % 5]
    D = int(rng.choice([40, 136, 1000, 2048, 3456, 4104, 8408]))
    if H <= 200 and D > 4096:
        D = 1000
    B = int(rng.choice([1, 2, 3, 5, 8, 17]))
    T = int(rng.choice([5, 33, 64, 97, 130]))
    if B * T * D > 30e6:
        B = max(1, int(30e6 // (T * D)))
    lengths = [int(rng.randint(1, T + 1)) for _ in range(B)]
    lengths[0] = T
    dtype = torch.bfloat16 if seed % 2 == 0 else torch.float16
    opts = [{}, {"den_pair": 1}, {"den_dma": 2}, {"den_segments": 2}, {"gamma16": 1}][int(rng.randint(5))]
    den = syn.make_den_graph(H, K, D, seed=seed)
    x = syn.make_input(B, T, D, seed=190 + seed, device=DEV).to(dtype)
    L = torch.tensor(lengths)
    num = syn.make_num_graphs([max(4, l) for l in lengths], D, seed=800 + seed) if min(lengths) >= 4 else None

    def both(fn):
        out = []
        ctx = [_lib.option(k, v) for k, v in opts.items()]
        for c in ctx:
            c.__enter__()
        try:
            for flag in (True, False):
                native.HALF_ROWS = flag
                try:
                    out.append(fn())
                finally:
                    native.HALF_ROWS = True
                torch.cuda.synchronize()
        finally:
            for c in reversed(ctx):
                c.__exit__()
        return out

    def den_step():
        xx = x.clone().requires_grad_(True)
        o = ChainFunction.apply(xx, L, ChainGraphBatch(den, B), 1e-5)
        o.backward()
        torch.cuda.synchronize()
        assert int(ChainFunction.last_bad_count.sum()) == 0
        return float(o.detach()), xx.grad
    (o1, g1), (o0, g0) = both(den_step)
    assert g1.dtype == dtype and o1 == o0 and torch.equal(g1, g0), (H, K, D, B, T, opts, str(dtype))
    if num is not None:
        def loss_step():
            xx = x.clone().requires_grad_(True)
            loss = ChainLoss(den, 1e-5)(xx, L, num)
            loss.backward()
            torch.cuda.synchronize()
            assert int(ChainFunction.last_bad_count.sum()) == 0
            return float(loss.detach()), xx.grad
        (l1, h1), (l0, h0) = both(loss_step)
        assert h1.dtype == dtype and l1 == l0 and torch.equal(h1, h0), (H, K, D, B, T, opts, str(dtype))


@pytest.mark.parametrize("seed", range(8))
def test_random_numerators_in_the_reference_arithmetic(seed):
    """Option num_compat against the fp32 flavour of the oracle - the reference's LogAdd arithmetic - on
    random branching numerator graphs: padded per-utterance batches and one shared graph, unreachable states, several arcs of
    one pdf into a state, network outputs far apart (the cut-off at work), ragged lengths; objective and gradient through
    ChainFunction."""
    from helpers import _rand_num_fst
    from pychain_amd import ChainGraph, native
    rs = np.random.RandomState(4000 + seed)
    D = int(rs.choice([12, 48, 300, 2048]))
    B = int(rs.randint(1, 5))
    T = int(rs.choice([7, 40, 150]))
    fin = lambda H: {H - 1: 0.0, max(0, H - 2): -0.3}
    shared = seed % 3 == 0
    if shared:
        g = ChainGraph(_rand_num_fst(rs, int(rs.randint(3, min(T, 90))), int(rs.randint(0, 60)), D, fin), log_domain=True)
        gb = ChainGraphBatch(g, B)
    else:
        gs = [ChainGraph(_rand_num_fst(rs, int(rs.randint(3, min(T, 90))), int(rs.randint(0, 60)), D, fin), log_domain=True) for _ in range(B)]
        gb = ChainGraphBatch(gs, max_num_transitions=max(g.num_transitions for g in gs), max_num_states=max(g.num_states for g in gs))
    lengths = sorted((int(rs.randint(max(3, gb.num_states), T + 1)) if T > gb.num_states else T for _ in range(B)), reverse=True)
    lengths[0] = T
    L = torch.tensor(lengths)
    x = syn.make_input(B, T, D, seed=300 + seed, scale=float(rs.choice([1.0, 2.5, 5.0])))
    ro, rlg, _ = orc.num(gb, x.clamp(-30, 30), L, shared=shared, flavour="f32")
    if not np.isfinite(ro).all():
        pytest.skip("this draw has an utterance its graph cannot produce")
    with _lib.option("num_compat", 1):
        xx = x.to(DEV).requires_grad_(True)
        o = ChainFunction.apply(xx, L, gb)
        o.backward()
        torch.cuda.synchronize()
        assert int(ChainFunction.last_bad_count.sum()) == 0
    assert abs(float(o.detach()) - float(ro.sum())) <= 3e-6 * abs(float(ro.sum())) + 1e-5
    rg = np.exp(rlg.astype(np.float64))
    assert rel_err(xx.grad.cpu().numpy(), rg) <= 2e-5, rel_err(xx.grad.cpu().numpy(), rg)


@pytest.mark.parametrize("seed", range(8))
def test_random_hub_graphs_states_on_several_lanes(seed):
    """Random graphs with hub states (helpers.random_hub_graph; the CPU suite emulates the same ten): the plan puts states on
    several lanes on both sides - or, where hubs feed hubs so much that the occupancy tile would repeat its arcs beyond 3/2 of
    them (plan.cpp: the gamma-arc bound; seed 1), on neither; the lazy recursions' NC form (four waves or sixteen), the two-barrier
    kernel and - with the pair kernel asked for, which does not take such plans - whatever runs instead, against the fp64 oracle
    and against the plan with every state on one lane; ragged lengths, exact zeros in the padding."""
    from helpers import random_hub_graph
    from pychain_amd import _plan
    den, D = random_hub_graph(seed)
    rng = np.random.RandomState(3000 + seed)
    B, T = int(rng.randint(2, 7)), int(rng.randint(20, 90))
    lengths = sorted((int(rng.randint(1, T + 1)) for _ in range(B)), reverse=True)
    lengths[0] = T
    x = syn.make_input(B, T, D, seed=70 + seed)
    L = torch.tensor(lengths)
    plan = _plan.graph_plan(den, D, torch.device(DEV))
    split = plan.num_states > den.num_states
    assert split == bool((plan.slot_rows >> 28) & 1) and (split or seed == 1)
    ro, rg = orc.chain_function(x, L, ChainGraphBatch(den, B), 1e-5, flavour="f64")
    with _lib.option("plan_split", "0"):
        o0, g0 = _run(x, L, den, 1, False)
    for name, lazy, pair in (("lazy", 1, False), ("two-barrier", 0, False), ("pair asked for", 1, True)):
        o, g = _run(x, L, den, lazy, pair)
        assert abs(o - ro) <= 1e-5 * abs(ro) and rel_err(g, rg) <= 2e-5, (name, seed, o, ro, rel_err(g, rg))
        assert abs(o - o0) <= 1e-5 * abs(o0) and rel_err(g, g0) <= 2e-5, (name, seed)
        for b, l in enumerate(lengths):
            assert np.all(g[b, l:] == 0.0), (name, b)

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
        o, g = run()
        assert o == o0 and torch.equal(g, g0), (H, K, D, B, T, lengths[:8])

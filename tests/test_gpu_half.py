"""bf16 / fp16 network outputs read by the kernels AS THEY ARE: 2-byte rows converted
where they land (LDS-direct loads of the lazy recursions, the pair recursion's and the occupancy kernels' register staging,
the numerator's row-staging waves, the rows exp'd ahead), the gradient rounded to the same type where it is written - no
up-cast pass, no fp32 copy of [B,T,D], no cast back.  The arithmetic is the fp32 path's operation for operation, so the
result must be BIT-IDENTICAL to up-casting on the host side of the ABI (native.HALF_ROWS = False: x.float(), fp32 kernels,
gradient .to(dtype))."""
import numpy as np
import pytest
import torch

import oracle as orc
from helpers import rel_err
from pychain_amd import ChainFunction, ChainGraphBatch, ChainLoss, _lib, _plan, native, synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _both_ways(fn):
    outs = []
    for flag in (True, False):
        native.HALF_ROWS = flag
        try:
            outs.append(fn())
        finally:
            native.HALF_ROWS = True
        torch.cuda.synchronize()
    return outs


def _loss_step(den, x, L, num, scale=1.0):
    xx = x.clone().requires_grad_(True)
    loss = ChainLoss(den, 1e-5)(xx, L, num)
    (loss * scale if scale != 1.0 else loss).backward()
    torch.cuda.synchronize()
    assert int(ChainFunction.last_bad_count.sum()) == 0
    return float(loss.detach()), xx.grad


def _fn_step(x, L, gb):
    xx = x.clone().requires_grad_(True)
    o = ChainFunction.apply(xx, L, gb, 1e-5) if not gb.log_domain else ChainFunction.apply(xx, L, gb)
    o.backward()
    torch.cuda.synchronize()
    assert int(ChainFunction.last_bad_count.sum()) == 0
    return float(o.detach()), xx.grad


CASES = [
    # name, H, K, D, T, lengths, options
    ("C3-lazy-dma", 3000, 30000, 3456, 130, [130, 129, 64, 5], {}),
    ("C2-small", 200, 2000, 1000, 150, [150, 77, 4, 150], {}),
    ("C1-small", 20, 60, 40, 50, [50, 37], {}),
    ("C3-pair", 3000, 30000, 3456, 96, [96, 95, 33, 96, 7], {"den_pair": 1}),
    ("C3-gated", 3000, 30000, 3456, 300, [300, 290, 100], {"den_segments": 3}),
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_two_byte_rows_are_the_up_cast_rows_bit_for_bit(case, dtype):
    name, H, K, D, T, lens, opts = case
    den = syn.make_den_graph(H, K, D, seed=0)
    L = torch.tensor(lens)
    B = len(lens)
    x = syn.make_input(B, T, D, seed=17, device=DEV).to(dtype)
    num = syn.make_num_graphs(lens, D, seed=300) if min(lens) >= 4 else None
    plan = _plan.graph_plan(den, D, torch.device(DEV))
    ctx = [_lib.option(k, v) for k, v in opts.items()]
    for c in ctx:
        c.__enter__()
    try:
        lib = _lib.lib()
        assert lib.pychain_hip_den_half_native(plan.stride, plan.slot_rows, den.num_states, D, B, T) == 1
        # the denominator alone (rows exp'd ahead where the call does that: fp32 rows written from 2-byte ones)
        (o1, g1), (o0, g0) = _both_ways(lambda: _fn_step(x, L, ChainGraphBatch(den, B)))
        assert g1.dtype == dtype and o1 == o0 and torch.equal(g1, g0), name
        ro, rg = orc.chain_function(x.float().cpu(), L, ChainGraphBatch(den, B), 1e-5)
        eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
        assert abs(o1 - ro) <= 1e-4 * abs(ro) and rel_err(g1.float().cpu().numpy(), rg) <= eps + 1e-4
        if num is not None:
            assert lib.pychain_hip_chain_loss_half_native(plan.stride, plan.slot_rows, den.num_states, D, B, T, num.num_states,
                                                          num.forward_transitions.shape[1]) == 1
            (l1, h1), (l0, h0) = _both_ways(lambda: _loss_step(den, x, L, num))
            assert h1.dtype == dtype and l1 == l0 and torch.equal(h1, h0), name
            # the numerator alone: 2-byte rows in, fp32 gradient cast by the caller
            (n1, k1), (n0, k0) = _both_ways(lambda: _fn_step(x, L, num))
            assert k1.dtype == dtype and n1 == n0 and torch.equal(k1, k0), name
            # an upstream gradient that is not 1: the 2-byte gradient is rescaled in place (one more rounding)
            l3, h3 = _loss_step(den, x, L, num, scale=3.0)
            assert l3 == l1 and rel_err(h3.float().cpu().numpy(), 3.0 * h1.float().cpu().numpy()) <= 2 * eps
    finally:
        for c in reversed(ctx):
            c.__exit__()


def test_two_byte_rows_of_8408_pdfs():
    """C4's shape (rows beyond 4096 pdfs: the LzDma map, the one-frame occupancy kernel, rows exp'd ahead), denominator."""
    den = syn.make_den_graph(3000, 30000, 8408, seed=0)
    L = torch.tensor([120, 119, 64])
    x = syn.make_input(3, 120, 8408, seed=3, device=DEV).to(torch.bfloat16)
    for opts in ({"den_dma": 3}, {"den_dma": 2}):          # (rows exp'd ahead - no longer a default of this map - and by the recursions)
        ctx = [_lib.option(k, v) for k, v in opts.items()]
        for c in ctx:
            c.__enter__()
        try:
            (o1, g1), (o0, g0) = _both_ways(lambda: _fn_step(x, L, ChainGraphBatch(den, 3)))
        finally:
            for c in reversed(ctx):
                c.__exit__()
        assert g1.dtype == torch.bfloat16 and o1 == o0 and torch.equal(g1, g0), opts


def test_no_fp32_copy_of_the_network_output_in_a_bf16_step():
    """The point of reading 2-byte rows in the kernels: a bf16 fused step allocates neither an fp32 copy of [B,T,D] nor an
    fp32 gradient (torch.cuda.max_memory_allocated); and a NaN in a bf16 row is still seen."""
    cfg = syn.CONFIGS["C3"]
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    B, T, D = 8, 256, cfg["D"]
    L = torch.tensor([256, 250, 200, 199, 130, 64, 33, 256])
    num = syn.make_num_graphs(L.tolist(), D, seed=400)
    x = syn.make_input(B, T, D, seed=19, device=DEV).to(torch.bfloat16)
    _loss_step(den, x, L, num)                                   # plans, workspaces, graph uploads: allocated
    peaks = []
    for flag in (True, False):
        native.HALF_ROWS = flag
        try:
            native.release_workspaces()
            torch.cuda.synchronize(); torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
            base = torch.cuda.memory_allocated()
            _loss_step(den, x, L, num)
            peaks.append(torch.cuda.max_memory_allocated() - base)
        finally:
            native.HALF_ROWS = True
    elts = B * T * D
    # up-cast: + x.float() (4 B / element) + an fp32 gradient (4 instead of 2) [+ the cast's output]
    assert peaks[0] + 6 * elts <= peaks[1] + (1 << 20), peaks
    xn = x.clone()
    xn[3, 17, 5] = float("nan")
    xx = xn.requires_grad_(True)
    loss = ChainLoss(den, 1e-5)(xx, L, num)
    torch.cuda.synchronize()
    assert int(ChainFunction.last_bad_count.sum()) > 0 and np.isnan(float(loss.detach()))


def test_shapes_without_two_byte_kernels_are_up_cast():
    """Rows that are not a multiple of 8 pdfs, the two-barrier recursion, a plan in the general format: the library says so
    (pychain_hip_*_half_native = 0, EUNSUPPORTED if called anyway) and the host side up-casts as before."""
    lib = _lib.lib()
    den = syn.make_den_graph(300, 3000, 1004, seed=2)              # 1004 = 4 * 251
    L = torch.tensor([40, 33])
    x = syn.make_input(2, 40, 1004, seed=4, device=DEV).to(torch.bfloat16)
    plan = _plan.graph_plan(den, 1004, torch.device(DEV))
    assert lib.pychain_hip_den_half_native(plan.stride, plan.slot_rows, den.num_states, 1004, 2, 40) == 0
    o, g = _fn_step(x, L, ChainGraphBatch(den, 2))
    ro, rg = orc.chain_function(x.float().cpu(), L, ChainGraphBatch(den, 2), 1e-5)
    assert g.dtype == torch.bfloat16 and abs(o - ro) <= 1e-4 * abs(ro) and rel_err(g.float().cpu().numpy(), rg) <= 2.0 ** -8 + 1e-4
    objf = torch.empty(2, device=DEV); bad = torch.empty(1, dtype=torch.int32, device=DEV)
    ws = torch.empty(lib.pychain_hip_den_workspace_min_bytes(2, 40, den.num_states, 1004), dtype=torch.uint8, device=DEV)
    gr = torch.empty_like(x)
    rc = lib.pychain_hip_den_forward_backward(plan.blob.data_ptr(), plan.stride, plan.slot_rows, den.num_states, 1004, x.data_ptr(),
                                              _lib.BF16, 0, L.to(DEV).data_ptr(), 2, 40, 1e-5, 1.0, objf.data_ptr(), gr.data_ptr(),
                                              bad.data_ptr(), 0, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == -2                                                # PYCHAIN_HIP_EUNSUPPORTED
    den8 = syn.make_den_graph(300, 3000, 1000, seed=2)
    plan8 = _plan.graph_plan(den8, 1000, torch.device(DEV))
    with _lib.option("den_lazy", 0):
        assert lib.pychain_hip_den_half_native(plan8.stride, plan8.slot_rows, den8.num_states, 1000, 2, 40) == 0
    assert lib.pychain_hip_den_half_native(plan8.stride, plan8.slot_rows, den8.num_states, 1000, 2, 40) == 1

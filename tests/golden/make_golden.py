#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference):

    python oracle/build_ref.py && python tests/golden/make_golden.py

What runs:
  * native numerics: oracle/_ref/pychain_C.so = the reference's own
    pytorch_binding/src/{pychain,base,chain-computation,chain-log-domain-computation}.cc,
    compiled unmodified (oracle/build_ref.py);
  * Python layer: the reference's own pychain/loss.py and pychain/graph.py,
    imported from /root/reference (ChainFunction, ChainLoss, ChainGraphBatch).

What does NOT run: the reference's `simplefst` (openfst_binding/, needs OpenFST which
this image lacks).  pychain/graph.py imports it at module scope, so a placeholder
module is registered to let that import succeed.  For G1-G5 / A5 reference ChainGraph
objects are created with `__new__` and populated with the tensors of this repo's
fst_to_tensor restatement.  For A4 (`gen_chaingraph_init`) the reference's REAL
`ChainGraph.__init__` (pychain/graph.py:25-70) is executed for every
(initial_mode, final_mode, log_domain) combination, with the placeholder's
`StdVectorFst` pointing at this repo's restatement for the three calls it makes
(fst_to_tensor, start_state, set_leaky_probs - SURVEY.md Appendix A.3): the mode
handling of the constructor is pinned by the reference, the FST->tensor layout is not
(it needs OpenFST; DESIGN.md says so).

Fixtures are data only: inputs + the reference's outputs (.npz).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(REPO, "oracle"))
import ref_loader  # noqa: E402

pychain_C = ref_loader.load()
sys.modules["pychain_C"] = pychain_C
sys.modules["simplefst"] = types.ModuleType("simplefst")  # import placeholder only, see docstring
sys.path.insert(0, "/root/reference")
import pychain as ref_pychain  # noqa: E402
assert ref_pychain.__file__.startswith("/root/reference/"), ref_pychain.__file__
from pychain.graph import ChainGraph as RefChainGraph, ChainGraphBatch as RefChainGraphBatch  # noqa: E402
from pychain.loss import ChainFunction as RefChainFunction, ChainLoss as RefChainLoss  # noqa: E402

sys.path.insert(0, REPO)
from pychain_amd import synthetic as syn  # noqa: E402
from pychain_amd.graph import ChainGraph  # noqa: E402
from pychain_amd.simplefst import StdVectorFst  # noqa: E402

torch.set_num_threads(1)
GRAPH_FIELDS = ["forward_transitions", "forward_transition_probs", "forward_transition_indices",
                "backward_transitions", "backward_transition_probs", "backward_transition_indices",
                "final_probs", "initial_probs", "leaky_probs"]


def to_ref_graph(g):
    """Reference ChainGraph object carrying this repo's graph tensors."""
    r = RefChainGraph.__new__(RefChainGraph)
    for f in GRAPH_FIELDS + ["num_states", "log_domain", "num_transitions", "is_empty", "start_state"]:
        v = getattr(g, f)
        setattr(r, f, v.clone() if torch.is_tensor(v) else v)
    return r


def graph_arrays(g, prefix):
    out = {}
    for f in GRAPH_FIELDS:
        v = getattr(g, f)
        if v is not None:
            out[prefix + f] = v.numpy()
    out[prefix + "start_state"] = np.int64(g.start_state)
    out[prefix + "log_domain"] = np.bool_(g.log_domain)
    return out


def batch_arrays(gb, prefix):
    out = {}
    for f in GRAPH_FIELDS + ["start_state"]:
        v = getattr(gb, f)
        if v is not None:
            out[prefix + f] = v.numpy()
    out[prefix + "num_states"] = np.int64(gb.num_states)
    out[prefix + "batch_size"] = np.int64(gb.batch_size)
    out[prefix + "log_domain"] = np.bool_(gb.log_domain)
    return out


def run_function(x, lengths, ref_batch, leaky=1e-5):
    xx = x.clone().requires_grad_(True)
    objf = RefChainFunction.apply(xx, lengths, ref_batch, leaky)
    objf.backward()
    return objf.detach().numpy().astype(np.float32), xx.grad.numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %7.1f KB" % (name, os.path.getsize(path) / 1024.0))


def small_num_graphs(lengths, D, seed, sizes):
    graphs = [ChainGraph(syn.make_num_fst(h, D, seed + i), log_domain=True) for i, h in enumerate(sizes)]
    max_k = max(g.num_transitions for g in graphs)
    max_h = max(g.num_states for g in graphs)
    return graphs, max_k, max_h


# ---------------------------------------------------------------- G1 / G2: C1
def gen_c1():
    cfg = syn.CONFIGS["C1"]
    B, T, H, K, D = cfg["B"], cfg["T"], cfg["H"], cfg["K"], cfg["D"]
    lengths = torch.tensor(cfg["lengths"], dtype=torch.long)
    den = syn.make_den_graph(H, K, D, seed=0)
    x = syn.make_input(B, T, D, seed=1)
    ref_den = to_ref_graph(den)
    objf, grad = run_function(x, lengths, RefChainGraphBatch(ref_den, B))
    save("g1_c1_den", x=x.numpy(), lengths=lengths.numpy(), leaky_coefficient=np.float32(1e-5),
         objf=objf, grad=grad, **graph_arrays(den, "den_"))

    graphs, max_k, max_h = small_num_graphs(lengths.tolist(), D, 100, [12, 7])
    ref_num = RefChainGraphBatch([to_ref_graph(g) for g in graphs], max_num_transitions=max_k,
                                 max_num_states=max_h)
    num_objf, num_grad = run_function(x, lengths, ref_num)
    out = dict(x=x.numpy(), lengths=lengths.numpy(), den_objf=objf, den_grad=grad,
               num_objf=num_objf, num_grad=num_grad, max_k=np.int64(max_k), max_h=np.int64(max_h))
    for avg in (True, False):
        xx = x.clone().requires_grad_(True)
        loss = RefChainLoss(ref_den, 1e-5, avg=avg)(xx, lengths, ref_num)
        loss.backward()
        out["loss_avg%d" % avg] = loss.detach().numpy()
        out["xgrad_avg%d" % avg] = xx.grad.numpy()
    out.update(graph_arrays(den, "den_"))
    for i, g in enumerate(graphs):
        out.update(graph_arrays(g, "num%d_" % i))
    out.update(batch_arrays(ref_num, "numbatch_"))
    save("g2_c1_chainloss", **out)


# ---------------------------------------------------------------- G3: variants
def gen_variants():
    D = 30
    # graph with: a state without in-arcs (3), a state without out-arcs (5), duplicate
    # arcs, unused pdfs (only pdfs < 20 used), non-trivial finals.
    arcs = [(0, 1, 2, -0.5), (0, 1, 2, -0.5), (0, 2, 7, -1.2), (1, 1, 3, -0.3), (1, 2, 4, -1.5),
            (2, 0, 5, -0.9), (2, 4, 6, -0.7), (3, 0, 8, -0.4), (3, 4, 9, -1.1), (4, 4, 10, -0.2),
            (4, 5, 11, -1.9), (4, 0, 12, -1.0), (1, 5, 13, -2.0), (2, 2, 19, -0.6), (0, 4, 0, -1.3)]
    finals = {0: -0.5, 2: -1.0, 4: -0.1, 5: 0.0}
    fst = StdVectorFst.from_arcs(6, 0, arcs, finals)
    rs = np.random.RandomState(7)
    cases = {
        "leaky_ones": dict(initial_mode="leaky", final_mode="ones", coef=1e-5, lengths=[23, 23, 9]),
        "fst_fst": dict(initial_mode="fst", final_mode="fst", coef=1e-5, lengths=[23, 17, 9]),
        "leaky_fst_coef01": dict(initial_mode="leaky", final_mode="fst", coef=0.1, lengths=[23, 17, 1]),
        "fst_ones_clamp": dict(initial_mode="fst", final_mode="ones", coef=1e-3, lengths=[23, 5, 2],
                               big=True),
    }
    out = {}
    for name, c in cases.items():
        g = ChainGraph(fst, initial_mode=c["initial_mode"], final_mode=c["final_mode"])
        lengths = torch.tensor(c["lengths"], dtype=torch.long)
        B, T = len(c["lengths"]), max(c["lengths"])
        x = torch.from_numpy(rs.randn(B, T, D).astype(np.float32) * (25.0 if c.get("big") else 3.0))
        objf, grad = run_function(x, lengths, RefChainGraphBatch(to_ref_graph(g), B), c["coef"])
        p = name + "__"
        out[p + "x"] = x.numpy(); out[p + "lengths"] = lengths.numpy()
        out[p + "coef"] = np.float32(c["coef"]); out[p + "objf"] = objf; out[p + "grad"] = grad
        out.update(graph_arrays(g, p + "den_"))
    save("g3_den_variants", **out)

    # numerator variants: branching log-domain graph with unreachable / dead-end states,
    # final_mode ones vs fst, inputs beyond the clamp.
    narcs = [(0, 0, 1, -0.7), (0, 1, 2, -0.7), (1, 1, 3, -0.4), (1, 2, 4, -1.1), (1, 3, 5, -2.0),
             (2, 2, 4, -0.6), (2, 3, 6, -0.8), (3, 3, 7, -0.3), (4, 3, 8, -0.5), (2, 5, 9, -1.5)]
    nfst = StdVectorFst.from_arcs(6, 0, narcs, {3: -0.2, 2: -1.7})
    out = {}
    for name, fm, big in (("fst", "fst", False), ("ones", "ones", False), ("clamp", "fst", True)):
        g = ChainGraph(nfst, final_mode=fm, log_domain=True)
        g2 = ChainGraph(syn.make_num_fst(4, D, 55), log_domain=True)
        lengths = torch.tensor([19, 11, 4], dtype=torch.long)
        x = torch.from_numpy(rs.randn(3, 19, D).astype(np.float32) * (25.0 if big else 3.0))
        gl = [g, g2, g2]
        mk = max(a.num_transitions for a in gl); mh = max(a.num_states for a in gl)
        rb = RefChainGraphBatch([to_ref_graph(a) for a in gl], max_num_transitions=mk, max_num_states=mh)
        objf, grad = run_function(x, lengths, rb)
        p = name + "__"
        out[p + "x"] = x.numpy(); out[p + "lengths"] = lengths.numpy()
        out[p + "objf"] = objf; out[p + "grad"] = grad
        out.update(batch_arrays(rb, p + "batch_"))
    save("g3_num_variants", **out)


# ---------------------------------------------------------------- G4: medium den
def gen_medium():
    B, T, H, K, D = 4, 300, 200, 2000, 1000
    lengths = torch.tensor([300, 271, 200, 163], dtype=torch.long)
    den = syn.make_den_graph(H, K, D, seed=3)
    x = syn.make_input(B, T, D, seed=4)
    objf, grad = run_function(x, lengths, RefChainGraphBatch(to_ref_graph(den), B))
    per_seq = []
    for b in range(B):
        o, _ = run_function(x[b:b + 1, :lengths[b]], lengths[b:b + 1], RefChainGraphBatch(to_ref_graph(den), 1))
        per_seq.append(o)
    rows = np.array([[0, 0], [0, 150], [0, 299], [1, 270], [2, 7], [2, 199], [3, 100], [3, 162]])
    w = syn.uniform(99, B * T * D).reshape(B, T, D)
    save("g4_den_medium", B=B, T=T, H=H, K=K, D=D, graph_seed=3, x_seed=4, lengths=lengths.numpy(),
         objf=objf, objf_per_seq=np.array(per_seq, dtype=np.float32),
         grad_rowsum=grad.astype(np.float64).sum(-1), sample_rows=rows,
         grad_rows=grad[rows[:, 0], rows[:, 1]], grad_checksum=(grad.astype(np.float64) * w).sum(),
         grad_absmax=np.abs(grad).max())

    # medium numerator (ragged, H_n ~ T/4)
    Bn, Dn = 4, 300
    nl = [300, 271, 200, 163]
    lengths = torch.tensor(nl, dtype=torch.long)
    gb = syn.make_num_graphs(nl, Dn, seed=200)
    x = syn.make_input(Bn, 300, Dn, seed=5)
    rb = RefChainGraphBatch.__new__(RefChainGraphBatch)
    for f in GRAPH_FIELDS + ["start_state", "num_states", "batch_size", "log_domain", "num_transitions"]:
        setattr(rb, f, getattr(gb, f))
    objf, grad = run_function(x, lengths, rb)
    w = syn.uniform(98, Bn * 300 * Dn).reshape(Bn, 300, Dn)
    save("g4_num_medium", B=Bn, T=300, D=Dn, graph_seed=200, x_seed=5, lengths=lengths.numpy(),
         objf=objf, grad_rowsum=grad.astype(np.float64).sum(-1), sample_rows=rows,
         grad_rows=grad[rows[:, 0], rows[:, 1]], grad_checksum=(grad.astype(np.float64) * w).sum())


# ---------------------------------------------------------------- G5: raw pychain_C level
def gen_raw():
    cfg = syn.CONFIGS["C1"]
    B, T, H, K, D = cfg["B"], cfg["T"], cfg["H"], cfg["K"], cfg["D"]
    lengths = torch.tensor(cfg["lengths"], dtype=torch.long)
    den = syn.make_den_graph(H, K, D, seed=0)
    x = syn.make_input(B, T, D, seed=1).clamp(-30, 30)
    db = RefChainGraphBatch(to_ref_graph(den), B)
    bs = torch.nn.utils.rnn.pack_padded_sequence(x, lengths, batch_first=True).batch_sizes
    objf, grad, ok = pychain_C.forward_backward(
        db.forward_transitions, db.forward_transition_indices, db.forward_transition_probs,
        db.backward_transitions, db.backward_transition_indices, db.backward_transition_probs,
        db.leaky_probs, db.initial_probs, db.final_probs, db.start_state, x.exp(), bs, lengths,
        db.num_states, 1e-5)
    graphs, mk, mh = small_num_graphs(lengths.tolist(), D, 100, [12, 7])
    nb = RefChainGraphBatch([to_ref_graph(g) for g in graphs], max_num_transitions=mk, max_num_states=mh)
    nobjf, nlg, nok = pychain_C.forward_backward_log_domain(
        nb.forward_transitions, nb.forward_transition_indices, nb.forward_transition_probs,
        nb.backward_transitions, nb.backward_transition_indices, nb.backward_transition_probs,
        nb.initial_probs, nb.final_probs, nb.start_state, x, bs, lengths, nb.num_states)
    save("g5_raw_pychain_C", x_clamped=x.numpy(), lengths=lengths.numpy(), batch_sizes=bs.numpy(),
         den_objf=objf.numpy(), den_grad=grad.numpy(), den_ok=ok.numpy(),
         num_objf=nobjf.numpy(), num_log_grad=nlg.numpy(), num_ok=nok.numpy(),
         **batch_arrays(db, "den_"), **batch_arrays(nb, "num_"))


# ---------------------------------------------------------------- A4/A5: container snapshots
def gen_containers():
    den = syn.make_den_graph(6, 14, 9, seed=11)
    rb = RefChainGraphBatch(to_ref_graph(den), 3)
    out = batch_arrays(rb, "byone_")
    out.update(graph_arrays(den, "den_"))
    graphs, mk, mh = small_num_graphs([9, 7, 5], 9, 300, [5, 3, 4])
    rl = RefChainGraphBatch([to_ref_graph(g) for g in graphs], max_num_transitions=mk + 2, max_num_states=mh + 1)
    out.update(batch_arrays(rl, "bylist_"))
    out["bylist_num_transitions"] = np.int64(rl.num_transitions)
    rl.reorder(torch.tensor([2, 0, 1]))
    out.update(batch_arrays(rl, "reordered_"))
    for i, g in enumerate(graphs):
        out.update(graph_arrays(g, "g%d_" % i))
    # prob-domain list batch (leaky_probs padded with 0)
    pg = [syn.make_den_graph(5, 9, 9, seed=21), syn.make_den_graph(3, 5, 9, seed=22)]
    rp = RefChainGraphBatch([to_ref_graph(g) for g in pg], max_num_transitions=9, max_num_states=5)
    out.update(batch_arrays(rp, "problist_"))
    for i, g in enumerate(pg):
        out.update(graph_arrays(g, "pg%d_" % i))
    save("a5_containers", **out)


# ---------------------------------------------------------------- A4: the reference's ChainGraph.__init__
def gen_chaingraph_init():
    # start state 2 (not 0), three final states with different weights, a state without out-arcs
    arcs = [(0, 1, 3, -0.4), (0, 0, 1, -1.1), (1, 3, 0, -0.7), (2, 0, 2, -0.2), (2, 1, 4, -1.6), (3, 3, 5, -0.9),
            (3, 4, 6, -0.5), (1, 4, 2, -1.2)]
    fst = StdVectorFst.from_arcs(5, 2, arcs, {4: -0.3, 3: -1.5, 1: 0.0})
    sys.modules["simplefst"].StdVectorFst = StdVectorFst        # what the reference constructor calls into
    out = {"arcs": np.array(arcs, dtype=np.float64), "num_states": np.int64(5), "start": np.int64(2),
           "final_states": np.array([4, 3, 1]), "final_weights": np.array([-0.3, -1.5, 0.0])}
    try:
        for log_domain in (False, True):
            for initial_mode in ("fst", "leaky"):
                for final_mode in ("fst", "ones"):
                    tag = "%s_%s_%d__" % (initial_mode, final_mode, int(log_domain))
                    try:
                        g = RefChainGraph(fst, initial_mode=initial_mode, final_mode=final_mode, log_domain=log_domain)
                    except AssertionError as e:
                        out[tag + "raises"] = np.array(str(e))
                        continue
                    out.update(graph_arrays(g, tag))
                    out[tag + "num_states"] = np.int64(g.num_states)
                    out[tag + "num_transitions"] = np.int64(g.num_transitions)
                    out[tag + "is_empty"] = np.bool_(g.is_empty)
                    out[tag + "leaky_is_none"] = np.bool_(g.leaky_probs is None)
        try:
            RefChainGraph(StdVectorFst.from_arcs(2, 0, [], {1: 0.0}))
            out["empty_raises"] = np.array("")
        except Exception as e:          # graph.py:70
            out["empty_raises"] = np.array(str(e))
    finally:
        del sys.modules["simplefst"].StdVectorFst
    save("a4_chaingraph_init", **out)


# ---------------------------------------------------------------- G6: benchmark-length sequences
def _ref_batch_from(gb):
    """Reference ChainGraphBatch carrying the tensors of a collated pychain_amd batch."""
    rb = RefChainGraphBatch.__new__(RefChainGraphBatch)
    for f in GRAPH_FIELDS + ["start_state", "num_states", "batch_size", "log_domain"]:
        v = getattr(gb, f)
        setattr(rb, f, v.clone() if torch.is_tensor(v) else v)
    rb.num_transitions = getattr(gb, "num_transitions", None)
    return rb


def gen_long(names=None, fixture="g6_long"):
    """G6: the long cases of tests/helpers.py:long_case through the REAL reference binary - at
    T >= 700 its fp32 log-domain numerator (LogAdd with the -15.94 cut-off, base.h:14-32, chained through
    chain-log-domain-computation.cc:137-158,256-266) is itself > 1e-4 from the same equations in fp64.  Stored per case:
    objf, per-frame gradient row sums, sampled full gradient rows (the rows where the reference is furthest from the fp64
    evaluation + evenly spread ones), the fp64 evaluation (oracle/chain_oracle.c, REAL=double) of the same rows and row
    sums, and the reference's distance from fp64 over the WHOLE gradient."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import helpers
    out = {}
    for name in (names or helpers.LONG_CASES):
        c = helpers.long_case(name)
        x, L, B = c["x"], c["lengths"], c["x"].shape[0]
        if c["kind"] == "den":
            objf, grad = run_function(x, L, RefChainGraphBatch(to_ref_graph(c["den"]), B), c["leaky"])
        elif c["kind"] == "num":
            if c["num_list"] is not None and len(c["num_list"]) == 1:
                rb = RefChainGraphBatch(to_ref_graph(c["num_list"][0]), B)
            else:
                rb = _ref_batch_from(c["num"])
            objf, grad = run_function(x, L, rb)
        else:
            gs = c["num_list"]
            rb = RefChainGraphBatch([to_ref_graph(g) for g in gs], max_num_transitions=max(g.num_transitions for g in gs),
                                    max_num_states=max(g.num_states for g in gs))
            xx = x.clone().requires_grad_(True)
            loss = RefChainLoss(to_ref_graph(c["den"]), c["leaky"], avg=True)(xx, L, rb)
            loss.backward()
            objf, grad = loss.detach().numpy().astype(np.float32), xx.grad.numpy()
        o64, g64 = helpers.long_case_oracle(c, "f64")
        o32, g32 = helpers.long_case_oracle(c, "f32")
        ref = grad.astype(np.float64)
        absmax = float(np.abs(ref).max())
        dist = np.abs(ref - g64).max(-1)                              # [B, T]
        ref_vs_f64 = float(dist.max() / np.abs(g64).max())
        nrows = (12 if x.shape[2] > 1024 else 24) if c["kind"] == "den" else (12 if c["kind"] != "num" and x.shape[2] > 1024 else 160)   # (dense rows of 4 - 34 KB)
        worst = np.dstack(np.unravel_index(np.argsort(dist, axis=None)[::-1][:nrows // 2], dist.shape))[0]
        live = [(b, t) for b in range(B) for t in range(int(L[b]))]
        spread = np.array(live)[np.linspace(0, len(live) - 1, nrows - len(worst)).astype(int)]
        rows = np.unique(np.concatenate([worst, spread]), axis=0)
        p = name + "__"
        out[p + "objf"] = np.float64(objf); out[p + "objf_f64"] = np.float64(o64)
        out[p + "rows"] = rows.astype(np.int32)
        out[p + "ref_rows"] = grad[rows[:, 0], rows[:, 1]]
        out[p + "f64_rows"] = g64[rows[:, 0], rows[:, 1]].astype(np.float32)     # (rounding: 6e-8 of a value)
        out[p + "ref_rowsum"] = ref.sum(-1); out[p + "f64_rowsum"] = g64.sum(-1)
        out[p + "ref_absmax"] = np.float64(absmax)
        out[p + "ref_vs_f64"] = np.float64(ref_vs_f64)
        out[p + "ref_vs_f64_rowsum"] = np.float64(np.abs(ref.sum(-1) - g64.sum(-1)).max())
        out[p + "restatement_f32_vs_ref"] = np.float64(np.abs(g32 - ref).max() / absmax)     # (what it was when generated)
        out[p + "x_checksum"] = np.float64(helpers.long_case_checksum(c))
        print("%-16s objf %.6f (f64 %.6f, restatement-f32 %.6f)  grad: reference vs f64 %.3e, restatement-f32 vs reference %.3e,"
              " restatement-f32 vs f64 %.3e; %d rows" % (name, float(objf), o64, o32, ref_vs_f64, out[p + "restatement_f32_vs_ref"],
                                                         np.abs(g32 - g64).max() / np.abs(g64).max(), len(rows)))
    save(fixture, **out)


def gen_sensitivity(fixture="g8_sensitivity"):
    """G8: how far does the REFERENCE move when its input moves by ONE ulp?  The long numerator cases of
    G6 through the real binary twice - on x and on x with every element moved to a neighbouring float (direction from a fixed
    random stream) - and the distance between the two reference gradients in the survey's metric.  If that distance is of
    the order of 1e-4, "within 1e-4 of the reference" is not a property an implementation can have at that length: the
    reference does not have it with respect to itself."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import helpers
    out = {}

    def run(c, x):
        L, B = c["lengths"], x.shape[0]
        if c["kind"] == "den":
            return run_function(x, L, RefChainGraphBatch(to_ref_graph(c["den"]), B), c["leaky"])
        if c["kind"] == "num":
            if c["num_list"] is not None and len(c["num_list"]) == 1:
                rb = RefChainGraphBatch(to_ref_graph(c["num_list"][0]), B)
            else:
                rb = _ref_batch_from(c["num"])
            return run_function(x, L, rb)
        gs = c["num_list"]
        rb = RefChainGraphBatch([to_ref_graph(g) for g in gs], max_num_transitions=max(g.num_transitions for g in gs),
                                max_num_states=max(g.num_states for g in gs))
        xx = x.clone().requires_grad_(True)
        loss = RefChainLoss(to_ref_graph(c["den"]), c["leaky"], avg=True)(xx, L, rb)
        loss.backward()
        return loss.detach().numpy().astype(np.float32), xx.grad.numpy()

    for name in ("c3_slice_num", "num_shared_T720", "fold_T751", "c3_slice_den"):
        c = helpers.long_case(name)
        x = c["x"]
        rs = np.random.RandomState(12345)
        up = rs.randint(0, 2, size=x.numel()).astype(bool).reshape(x.shape)
        xn = x.numpy()
        xp = torch.from_numpy(np.where(up, np.nextafter(xn, np.float32(np.inf)), np.nextafter(xn, np.float32(-np.inf))).astype(np.float32))
        o0, g0 = run(c, x)
        o1, g1 = run(c, xp)
        g0 = g0.astype(np.float64); g1 = g1.astype(np.float64)
        d = float(np.abs(g0 - g1).max() / np.abs(g0).max())
        # the same question to the fp64 evaluation: what part of it is the true derivative of the gradient, what part rounding
        c2 = dict(c); c2["x"] = xp
        _, h0 = helpers.long_case_oracle(c, "f64")
        _, h1 = helpers.long_case_oracle(c2, "f64")
        d64 = float(np.abs(h0 - h1).max() / np.abs(h0).max())
        p = name + "__"
        out[p + "ref_vs_ref_1ulp"] = np.float64(d)
        out[p + "f64_vs_f64_1ulp"] = np.float64(d64)
        out[p + "objf"] = np.float64(o0); out[p + "objf_1ulp"] = np.float64(o1)
        print("%-16s reference(x) vs reference(x +- 1 ulp): grad %.3e (fp64 evaluation: %.3e), objf %.9g vs %.9g (rel %.2e)"
              % (name, d, d64, float(o0), float(o1), abs(float(o0) - float(o1)) / abs(float(o0))))
    save(fixture, **out)


if __name__ == "__main__":
    if "--only-g8" in sys.argv:
        gen_sensitivity()
        sys.exit(0)
    if "--only-g6" in sys.argv:
        gen_long()
        sys.exit(0)
    if "--only-g7" in sys.argv:                    # (round 4: the C4 shape through the real binary; G6 is left as it is)
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import helpers as _h
        gen_long(_h.LONG_CASES_G7, "g7_c4_slice")
        sys.exit(0)
    gen_chaingraph_init()
    if "--only-a4" in sys.argv:
        sys.exit(0)
    gen_long()
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import helpers as _h
    gen_long(_h.LONG_CASES_G7, "g7_c4_slice")
    gen_c1()
    gen_variants()
    gen_medium()
    gen_raw()
    gen_containers()

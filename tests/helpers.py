"""Shared helpers for the parity tests."""
import json
import os

import numpy as np
import torch

from pychain_amd.graph import ChainGraph, ChainGraphBatch

GRAPH_FIELDS = ["forward_transitions", "forward_transition_probs", "forward_transition_indices",
                "backward_transitions", "backward_transition_probs", "backward_transition_indices",
                "final_probs", "initial_probs", "leaky_probs"]


def graph_from_npz(z, prefix):
    """Rebuild a ChainGraph from the arrays `graph_arrays` stored in a fixture."""
    t = {f: torch.from_numpy(np.array(z[prefix + f])) for f in GRAPH_FIELDS if prefix + f in z.files}
    return ChainGraph.from_tensors(
        t["forward_transitions"], t["forward_transition_probs"], t["forward_transition_indices"],
        t["backward_transitions"], t["backward_transition_probs"], t["backward_transition_indices"],
        t["final_probs"], t["initial_probs"], t.get("leaky_probs"),
        start_state=int(z[prefix + "start_state"]), log_domain=bool(z[prefix + "log_domain"]))


class RawBatch(object):
    """A ChainGraphBatch-like bag of batched tensors read from a fixture."""
    shared_graph = None

    def __init__(self, z, prefix):
        for f in GRAPH_FIELDS + ["start_state"]:
            if prefix + f in z.files:
                setattr(self, f, torch.from_numpy(np.array(z[prefix + f])))
            else:
                setattr(self, f, None)
        self.num_states = int(z[prefix + "num_states"])
        self.batch_size = int(z[prefix + "batch_size"])
        self.log_domain = bool(z[prefix + "log_domain"])


def batch_from_npz(z, prefix):
    rb = RawBatch(z, prefix)
    gb = ChainGraphBatch.__new__(ChainGraphBatch)
    gb.__dict__.update(rb.__dict__)
    gb.shared_graph = None
    gb._device_cache = {}
    return gb


def rel_err(a, b):
    """max |a-b| / max |b|  (the survey's gradient metric)."""
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    d = np.abs(a - b).max()
    return d / max(np.abs(b).max(), 1e-30)


class OracleChainLoss(torch.nn.Module):
    """ChainLoss(avg=False) evaluated by the CPU oracle, with autograd: the TEST stand-in for the per-rank loss where
    there is no GPU (tests/test_parallel.py, examples/train_tdnn.py --loss-cls helpers:OracleChainLoss).  What is
    under test there is the collective / DDP wiring around it."""

    def __init__(self, den_graph, leaky, avg=False):
        super().__init__()
        self.den_graph, self.leaky = den_graph, leaky

    def forward(self, x, lengths, num_graphs):
        import oracle as orc

        class F(torch.autograd.Function):
            @staticmethod
            def forward(ctx, xx):
                loss, grad = orc.chain_loss(xx, lengths, self.den_graph, num_graphs, self.leaky, avg=False)
                ctx.save_for_backward(torch.from_numpy(np.asarray(grad, dtype=np.float32)))
                return torch.tensor(float(loss))

            @staticmethod
            def backward(ctx, g):
                return ctx.saved_tensors[0] * g
        return F.apply(x)


def record_parity(name, **values):
    """Append the distances a parity test MEASURED to gpurun_out/parity_measured.jsonl (the summary committed under
    profiles/ comes from there): every literal bound in the tests has a measured number beside it."""
    root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "parity_measured.jsonl"), "a") as f:
            f.write(json.dumps(dict(test=name, **{k: float(v) for k, v in values.items()})) + "\n")
    except OSError:
        pass


# ---- the oracle fanned out over the host cores (full-size configurations) -------------------------------------
def _fanout_worker(rank, nworkers, x_cpu, lengths, den_graph, num_graphs, grad_hip, flavours, out_q):
    """Utterances rank, rank + nworkers, ...: den (+ num) by the CPU oracle in every flavour, compared HERE with the
    HIP gradient (shared memory) - only scalars travel back."""
    import oracle as orc
    from pychain_amd.graph import ChainGraphBatch
    torch.set_num_threads(1)
    res = []
    for b in range(rank, int(lengths.numel()), nworkers):
        L = int(lengths[b])
        xb = x_cpu[b:b + 1, :L].contiguous()
        lb = lengths[b:b + 1]
        den_b = ChainGraphBatch(den_graph, 1)
        num_b = None
        if num_graphs is not None:
            num_b = ChainGraphBatch.__new__(ChainGraphBatch)
            num_b.__dict__.update(num_graphs.__dict__)
            num_b._device_cache = {}
            num_b.reorder(torch.tensor([b]))
            num_b.batch_size = 1
        gh = grad_hip[b, :L].numpy().astype(np.float64)
        row = dict(b=b)
        for fl in flavours:
            d_o, d_g = orc.chain_function(xb, lb, den_b, 1e-5, flavour=fl)
            g = d_g[0].astype(np.float64)
            n_o = 0.0
            if num_b is not None:
                n_o, n_g = orc.chain_function(xb, lb, num_b, flavour=fl)
                g = g - n_g[0].astype(np.float64)
            row[fl] = dict(den_objf=float(d_o), num_objf=float(n_o), max_diff=float(np.abs(gh - g).max()),
                           max_ref=float(np.abs(g).max()))
            row["grad_" + fl] = g
        if "grad_f32" in row and "grad_f64" in row:       # the reference's own fp32 rounding: its distance from fp64
            row["f32"]["own_diff"] = float(np.abs(row["grad_f32"] - row["grad_f64"]).max())
        for fl in flavours:
            del row["grad_" + fl]
        row["tail_zero"] = bool((grad_hip[b, L:] == 0).all())
        res.append(row)
    out_q.put(res)


def oracle_fanout(x_cpu, lengths, den_graph, num_graphs, grad_hip, flavours=("f32", "f64")):
    """Per-utterance oracle results for a whole batch, one single-threaded worker process per host core (at most one
    per utterance).  Returns the list of per-utterance dicts of _fanout_worker, ordered by utterance."""
    import torch.multiprocessing as mp
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    nw = max(1, min(cores, int(lengths.numel())))
    import copy
    x_cpu = x_cpu.detach().float().cpu().contiguous().share_memory_()
    grad_hip = grad_hip.detach().float().cpu().contiguous().share_memory_()
    den_graph = copy.copy(den_graph)                   # without the device-resident plan / uploads cached on the objects
    den_graph._plan_cache = {}
    if num_graphs is not None:
        num_graphs = copy.copy(num_graphs)
        num_graphs._device_cache = {}
    ctx = mp.get_context("spawn")                      # (fork after HIP initialisation is unsafe)
    q = ctx.Queue()
    procs = [ctx.Process(target=_fanout_worker, args=(r, nw, x_cpu, lengths.cpu(), den_graph, num_graphs, grad_hip,
                                                      tuple(flavours), q)) for r in range(nw)]
    for p_ in procs:
        p_.start()
    import time
    rows, got, deadline = [], 0, time.time() + 1500
    while got < nw:
        try:
            rows += q.get(timeout=2)
            got += 1
        except Exception:
            if time.time() > deadline or any(p_.exitcode not in (None, 0) for p_ in procs):
                for p_ in procs:
                    p_.terminate()          # (our own children, by handle)
                raise RuntimeError("oracle worker failed or timed out")
    for p_ in procs:
        p_.join()
    return sorted(rows, key=lambda r: r["b"])


# ---- G6: the long-sequence cases pinned by the REAL reference binary (tests/golden/g6_long.npz) -----------------------
# The inputs are rebuilt from seeds by these builders on both sides (tests/golden/make_golden.py:gen_long runs the
# reference on them here; the tests run the oracle restatement / the HIP path on them), so the fixture holds only the
# reference's outputs.  Lengths where the reference's fp32 log-domain recursion (chain-log-domain-computation.cc:137-158,
# 256-266 over base.h:14-32) is itself > 1e-4 from the same equations in fp64.
LONG_CASES = ("c3_slice_den", "c3_slice_num", "num_shared_T720", "fold_T751")
LONG_CASES_G7 = ("c4_slice_den", "c2_slice_den")     # fixture g7_c4_slice.npz (round 4): rows of 8408 pdfs; a small graph


def _rand_num_fst(rs, H, extra, D, finals):
    """Left-to-right numerator graph with `extra` skip arcs (the branching graphs of the parity tests)."""
    from pychain_amd.simplefst import StdVectorFst
    arcs = [(s, s, int(rs.randint(D)), -0.5) for s in range(H)]
    arcs += [(s, s + 1, int(rs.randint(D)), -0.9) for s in range(H - 1)]
    arcs += [(int(a), int(min(H - 1, a + 1 + rs.randint(3))), int(rs.randint(D)), -1.3)
             for a in rs.randint(0, H - 1, size=extra)]
    arcs.sort(key=lambda a: a[0])
    return StdVectorFst.from_arcs(H, 0, arcs, finals(H))


def long_case(name):
    """dict(x [B,T,D] cpu f32, lengths, kind = "den" | "num" | "loss", den = ChainGraph | None,
    num = ChainGraphBatch | None, num_list = [ChainGraph] | None (the graphs `num` was collated from), leaky)."""
    from pychain_amd import synthetic as syn
    if name in ("c3_slice_den", "c3_slice_num"):
        # the C3 graph and pdf count, 4 ragged utterances up to the benchmark length (BASELINE.json configs[2])
        cfg = syn.CONFIGS["C3"]
        L = torch.tensor([1500, 1201, 977, 902])
        x = syn.make_input(4, 1500, cfg["D"], seed=1)
        if name == "c3_slice_den":
            return dict(x=x, lengths=L, kind="den", den=syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0),
                        num=None, num_list=None, leaky=1e-5)
        return dict(x=x, lengths=L, kind="num", den=None, num=syn.make_num_graphs(L.tolist(), cfg["D"], seed=100),
                    num_list=None, leaky=1e-5)
    if name == "c4_slice_den":
        # the C4 graph and pdf count (BASELINE.json configs[3]: rows of 8408 pdfs - the 16-wave map for rows beyond 4096 pdfs and
        # the rows exp'd ahead of the recursions), two utterances of 400 and 333 frames
        cfg = syn.CONFIGS["C4"]
        return dict(x=syn.make_input(2, 400, cfg["D"], seed=4), lengths=torch.tensor([400, 333]), kind="den",
                    den=syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0), num=None, num_list=None, leaky=1e-5)
    if name == "c2_slice_den":
        # the C2 graph (BASELINE.json configs[1]: 200 states - four-wave workgroups), three ragged utterances of the benchmark length
        cfg = syn.CONFIGS["C2"]
        return dict(x=syn.make_input(3, 150, cfg["D"], seed=6), lengths=torch.tensor([150, 149, 64]), kind="den",
                    den=syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0), num=None, num_list=None, leaky=1e-5)
    if name == "num_shared_T720":
        # one 700-state branching numerator graph shared by both utterances (graph stride 0), D = 48
        rs = np.random.RandomState(5)
        D = 48
        fin = lambda H: {H - 1: 0.0, H - 2: -0.4}
        _rand_num_fst(rs, 12, 20, D, fin); _rand_num_fst(rs, 9, 15, D, fin)     # (the stream position of the parity test)
        big = ChainGraph(_rand_num_fst(rs, 700, 300, D, fin), log_domain=True)
        return dict(x=syn.make_input(2, 720, D, seed=14), lengths=torch.tensor([720, 705]), kind="num", den=None,
                    num=ChainGraphBatch(big, 2), num_list=[big], leaky=1e-5)
    if name == "fold_T751":
        # fused ChainLoss: > 1024 distinct numerator pdfs per sequence, > 512 numerator states, branching, odd T
        rs = np.random.RandomState(11)
        D = 2048
        fin = lambda H: {H - 1: 0.0}
        gs = [ChainGraph(_rand_num_fst(rs, 700, 500, D, fin), log_domain=True),
              ChainGraph(_rand_num_fst(rs, 610, 40, D, fin), log_domain=True),
              ChainGraph(_rand_num_fst(rs, 30, 10, D, fin), log_domain=True)]
        gb = ChainGraphBatch(gs, max_num_transitions=max(g.num_transitions for g in gs),
                             max_num_states=max(g.num_states for g in gs))
        return dict(x=syn.make_input(3, 751, D, seed=29), lengths=torch.tensor([751, 640, 45]), kind="loss",
                    den=syn.make_den_graph(120, 900, D, seed=7), num=gb, num_list=gs, leaky=1e-5)
    raise KeyError(name)


def long_case_checksum(case):
    """One float64 over every input of a long case: the network output, the lengths and every graph tensor."""
    def cs(a):
        a = np.asarray(a, dtype=np.float64).ravel()
        a = np.where(np.isfinite(a), a, -77.0)
        return float((a * np.cos(np.arange(a.size, dtype=np.float64))).sum())
    tot = cs(case["x"].numpy()) + cs(case["lengths"].numpy())
    for g in (case["den"], case["num"]):
        if g is None:
            continue
        for f in GRAPH_FIELDS:
            v = getattr(g, f, None)
            if v is not None:
                tot += cs(v.numpy())
    return tot


def long_case_oracle(case, flavour):
    """(objf or loss, grad [B,T,D] float64) of a long case by the oracle restatement (oracle/chain_oracle.c)."""
    import oracle as orc
    B = case["x"].shape[0]
    if case["kind"] == "den":
        o, g = orc.chain_function(case["x"], case["lengths"], ChainGraphBatch(case["den"], B), case["leaky"], flavour=flavour)
    elif case["kind"] == "num":
        o, g = orc.chain_function(case["x"], case["lengths"], case["num"], flavour=flavour)
    else:
        o, g = orc.chain_loss(case["x"], case["lengths"], case["den"], case["num"], case["leaky"], avg=True, flavour=flavour)
    return float(o), np.asarray(g, dtype=np.float64)


class G6Case(object):
    """One case of tests/golden/g6_long.npz: the REAL reference's objective, per-frame gradient row sums and sampled
    gradient rows (the rows where it is furthest from the fp64 evaluation among them), the fp64 evaluation of the same
    rows, and the reference's own measured distance from fp64 - the yardstick of every comparison with it."""

    def __init__(self, z, name):
        p = name + "__"
        self.name = name
        self.objf = float(z[p + "objf"])
        self.objf_f64 = float(z[p + "objf_f64"])
        self.rows = np.array(z[p + "rows"])                    # [n, 2] (b, t)
        self.ref_rows = np.array(z[p + "ref_rows"], dtype=np.float64)
        self.f64_rows = np.array(z[p + "f64_rows"], dtype=np.float64)
        self.ref_rowsum = np.array(z[p + "ref_rowsum"])        # [B, T] float64
        self.f64_rowsum = np.array(z[p + "f64_rowsum"])
        self.ref_absmax = float(z[p + "ref_absmax"])           # max |grad| of the reference over the whole batch
        self.ref_vs_f64 = float(z[p + "ref_vs_f64"])           # max |ref - f64| / max |f64| over the WHOLE gradient
        self.ref_vs_f64_rowsum = float(z[p + "ref_vs_f64_rowsum"])
        self.x_checksum = float(z[p + "x_checksum"])

    def check_input(self, case):
        """the builder gave the bytes the fixture was generated from"""
        got = long_case_checksum(case)
        assert abs(got - self.x_checksum) <= 1e-9 * max(1.0, abs(self.x_checksum)), (got, self.x_checksum)

    def dist_ref(self, grad):
        """max |grad - reference| over the sampled rows / max |reference grad| (the survey's gradient metric)"""
        g = np.asarray(grad)[self.rows[:, 0], self.rows[:, 1]].astype(np.float64)
        return float(np.abs(g - self.ref_rows).max() / self.ref_absmax)

    def dist_f64(self, grad):
        g = np.asarray(grad)[self.rows[:, 0], self.rows[:, 1]].astype(np.float64)
        return float(np.abs(g - self.f64_rows).max() / self.ref_absmax)

    def dist_rowsum_ref(self, grad):
        return float(np.abs(np.asarray(grad, dtype=np.float64).sum(-1) - self.ref_rowsum).max())


def random_hub_graph(seed):
    """A random probability-domain graph with a few hub states - many arcs in, many arcs out, self-loops on hubs, hubs
    feeding hubs, final probabilities from the FST: what makes the plan compiler put states on several lanes
    (csrc/plan.cpp).  Returns (ChainGraph, num_pdfs)."""
    from pychain_amd import ChainGraph
    from pychain_amd.simplefst import StdVectorFst
    rs = np.random.RandomState(100 + seed)
    H = int(rs.randint(60, 420))
    D = int(rs.randint(20, 300))
    nh = int(rs.randint(1, 6))
    arcs = [(s, (s + 1) % H, int(rs.randint(D)), float(-0.1 - 2 * rs.rand())) for s in range(H)]
    arcs += [(int(rs.randint(H)), int(rs.randint(H)), int(rs.randint(D)), float(-0.1 - 2 * rs.rand())) for _ in range(int(rs.randint(3, 9)) * H)]
    hubs = rs.choice(H, size=nh, replace=False)
    for hb in hubs:
        fan_in, fan_out = int(rs.randint(20, 90)), int(rs.randint(0, 60))
        arcs += [(int(rs.randint(H)), int(hb), int(rs.randint(D)), float(-0.1 - 2 * rs.rand())) for _ in range(fan_in)]
        arcs += [(int(hb), int(rs.randint(H)), int(rs.randint(D)), float(-0.1 - 2 * rs.rand())) for _ in range(fan_out)]
        arcs += [(int(hb), int(hb), int(rs.randint(D)), -0.7), (int(hb), int(hubs[0]), int(rs.randint(D)), -1.1)]
    arcs.sort(key=lambda a: a[0])
    g = ChainGraph(StdVectorFst.from_arcs(H, 0, arcs, {s: float(-rs.rand()) for s in range(H)}), initial_mode="leaky", final_mode="fst")
    return g, D

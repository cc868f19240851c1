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

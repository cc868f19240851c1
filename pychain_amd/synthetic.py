"""Deterministic synthetic workloads for the LF-MMI path.

The reference ships no graphs, data or benchmarks, so bench.py, the tests and
smoke() all draw their inputs from here.  Everything is derived from a
splitmix64 counter stream -> identical bytes on every machine / numpy version.

Configs (BASELINE.json `configs`):
  C1  B=2,  T=50,   H=20,   K=60,    D=40     (reference CPU-runnable case)
  C2  B=64, T=150,  H=200,  K~2000,  D=1000
  C3  B=64, T<=1500, H=3000, K=30000, D=3456  + per-utterance numerator graphs
  C4  B=32, T=2000, H=3000, K=30000, D=8408
"""
import numpy as np
import torch

from .graph import ChainGraph, ChainGraphBatch
from .simplefst import StdVectorFst

__all__ = ["CONFIGS", "uniform", "normal", "make_den_fst", "make_den_graph", "make_structured_den_graph", "make_num_fst",
           "make_num_graphs", "make_lengths", "make_input", "make_input_utterances", "make_global_workload",
           "make_workload"]

CONFIGS = {
    "C1": dict(B=2, T=50, H=20, K=60, D=40, lengths=[50, 37], num=True),
    "C2": dict(B=64, T=150, H=200, K=2000, D=1000, lengths="equal", num=False),
    "C3": dict(B=64, T=1500, H=3000, K=30000, D=3456, lengths="ragged", num=True),
    "C4": dict(B=32, T=2000, H=3000, K=30000, D=8408, lengths="equal", num=False),
}

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(seed, n):
    """n uint64 values of the splitmix64 stream started at `seed` (vectorised:
    the state sequence is seed + (i+1)*gamma, so every output is independent)."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + (np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(seed, n):
    """float64 in [0,1)."""
    return (_splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def normal(seed, n):
    """Box-Muller on two independent streams; float64 N(0,1)."""
    u1 = uniform(seed * 2 + 1, n)
    u2 = uniform(seed * 2 + 2, n)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def randint(seed, n, hi):
    return (_splitmix64(seed, n) % np.uint64(hi)).astype(np.int64)


# ---------------------------------------------------------------------------
def make_den_fst(H, K, D, seed=0):
    """'Phone-trigram-like' denominator FST: a ring s->s+1 for connectivity plus
    K-H random arcs; pdf uniform in [0,D); arc log-prob uniform in [-2.1,-0.1];
    every state final with weight 0."""
    assert K >= H
    ring_src = np.arange(H, dtype=np.int64)
    ring_dst = (ring_src + 1) % H
    extra = K - H
    src = np.concatenate([ring_src, randint(seed * 10 + 1, extra, H)])
    dst = np.concatenate([ring_dst, randint(seed * 10 + 2, extra, H)])
    pdf = randint(seed * 10 + 3, K, D)
    lp = -2.1 + 2.0 * uniform(seed * 10 + 4, K)
    order = np.argsort(src, kind="stable")
    fin = np.zeros(H, dtype=np.float64)
    return StdVectorFst.from_arrays(H, 0, src[order], dst[order], pdf[order], lp[order], fin)


def make_den_graph(H, K, D, seed=0, initial_mode="leaky", final_mode="ones"):
    return ChainGraph(make_den_fst(H, K, D, seed), initial_mode=initial_mode,
                      final_mode=final_mode, log_domain=False)


def make_structured_den_graph(n_phone_inst=1500, fanout=9, D=3456, seed=7, loop_lp=-0.35, initial_mode="leaky", final_mode="ones"):
    """A phone-LM-like denominator (what composing a phone LM with the two-state chain topology gives; the benchmark graph
    of make_den_fst is random): every 'phone instance' is an entry state a (forward pdf) and a loop state b (self-loop
    pdf); both leave to the entry states of `fanout` successor instances.  EVERY ARC ENTERING A STATE CARRIES THAT STATE'S
    PDF, self-loops are strong (log-prob `loop_lp`).  H = 2 n states, K = n (2 + 2 fanout) arcs: 3000 / 30 000 by default."""
    n = n_phone_inst
    succ = randint(seed * 10 + 1, n * fanout, n).reshape(n, fanout)
    pdf_fwd = randint(seed * 10 + 2, n, D)
    pdf_loop = randint(seed * 10 + 3, n, D)
    lp_exit = -2.1 + 2.0 * uniform(seed * 10 + 4, n * fanout).reshape(n, fanout)
    a = 2 * np.arange(n, dtype=np.int64)
    b = a + 1
    # per instance: a -> b and b -> b (loop pdf), then a -> entry(s_j), b -> entry(s_j) for its successors (forward pdf of s_j)
    src = np.concatenate([a, b, np.repeat(a, fanout), np.repeat(b, fanout)])
    dst = np.concatenate([b, b, 2 * succ.reshape(-1), 2 * succ.reshape(-1)])
    pdf = np.concatenate([pdf_loop, pdf_loop, pdf_fwd[succ.reshape(-1)], pdf_fwd[succ.reshape(-1)]])
    lp = np.concatenate([np.full(n, loop_lp), np.full(n, loop_lp), lp_exit.reshape(-1) - 1.2, lp_exit.reshape(-1) - 1.2])
    order = np.argsort(src, kind="stable")
    fst = StdVectorFst.from_arrays(2 * n, 0, src[order], dst[order], pdf[order], lp[order], np.zeros(2 * n))
    return ChainGraph(fst, initial_mode=initial_mode, final_mode=final_mode, log_domain=False)


def make_num_fst(num_states, D, seed):
    """Left-to-right HMM: self-loop + forward arc per state, random pdfs,
    log-prob -0.7, last state final."""
    Hn = int(num_states)
    pdf_self = randint(seed * 10 + 5, Hn, D)
    pdf_fwd = randint(seed * 10 + 6, Hn, D)
    arcs = []
    for s in range(Hn):
        arcs.append((s, s, int(pdf_self[s]), -0.7))
        if s + 1 < Hn:
            arcs.append((s, s + 1, int(pdf_fwd[s]), -0.7))
    return StdVectorFst.from_arcs(Hn, 0, arcs, {Hn - 1: 0.0})


def make_num_graphs(lengths, D, seed=100, max_states=400):
    """One log-domain numerator graph per utterance, H_n = clamp(round(T_b/4), 4, max_states)
    (<= T_b so the numerator is finite), collated like pychain_example does."""
    graphs = []
    for b, Tb in enumerate(lengths):
        Hn = int(min(max(round(Tb / 4.0), 4), max_states, Tb))
        graphs.append(ChainGraph(make_num_fst(Hn, D, seed + b), log_domain=True))
    max_k = max(g.num_transitions for g in graphs)
    max_h = max(g.num_states for g in graphs)
    return ChainGraphBatch(graphs, max_num_transitions=max_k, max_num_states=max_h)


def make_lengths(B, T, mode, seed=2):
    if isinstance(mode, (list, tuple)):
        return torch.tensor(list(mode), dtype=torch.long)
    if mode == "equal":
        return torch.full((B,), T, dtype=torch.long)
    u = uniform(seed, B)
    L = np.floor(0.6 * T + u * (0.4 * T + 1)).astype(np.int64).clip(1, T)
    L[0] = T
    L = np.sort(L)[::-1].copy()
    return torch.from_numpy(L)


def make_input(B, T, D, seed=1, scale=2.0, device="cpu"):
    """[B,T,D] fp32, N(0,1)*scale."""
    if str(device) != "cpu":
        # Same distribution, generated on-device for the big configs (the byte-exact
        # host stream would take seconds per GB); parity checks at this size compare
        # HIP and oracle on the SAME tensor, so only reproducibility per device matters.
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        return torch.randn(B, T, D, generator=g, device=device, dtype=torch.float32) * scale
    x = normal(seed, B * T * D).astype(np.float32) * np.float32(scale)
    return torch.from_numpy(x.reshape(B, T, D))


def make_input_utterances(indices, T, D, seed=1, scale=2.0, device="cpu"):
    """Rows `indices` of a GLOBAL [B,T,D] network output that is never materialised: utterance i is its own random stream
    (seed, i), so every rank of a sharded run draws exactly the utterances it owns and all ranks agree on what utterance
    i is (bench.py --gpus N; a data loader reading utterances by index does the same)."""
    idx = [int(i) for i in indices]
    out = torch.empty(len(idx), T, D, dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    for j, i in enumerate(idx):
        g.manual_seed(int(seed) * 1000003 + i)
        out[j] = torch.randn(T, D, generator=g, device=device, dtype=torch.float32)
    return out.mul_(scale)


def make_global_workload(name, world_size, seed=0):
    """The global minibatch of a `world_size`-GPU run of BASELINE config `name`: world_size x B utterances (BASELINE.json's
    C5 = C3 at world_size 8: global B = 512) - lengths, the shared denominator graph and all numerator graphs, on the host,
    identical on every rank.  The network output is drawn per utterance (make_input_utterances)."""
    cfg = dict(CONFIGS[name])
    cfg["B_global"] = cfg["B"] * int(world_size)
    mode = cfg["lengths"]
    if isinstance(mode, (list, tuple)):
        mode = sorted(list(mode) * int(world_size), reverse=True)
    lengths = make_lengths(cfg["B_global"], cfg["T"], mode, seed=seed + 2)
    den = make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=seed)
    num = make_num_graphs(lengths.tolist(), cfg["D"], seed=seed + 100) if cfg["num"] else None
    return dict(lengths=lengths, den_graph=den, num_graphs=num, cfg=cfg)


def make_workload(name, device="cpu", seed=0, data_seed=None):
    """Returns dict(x, lengths, den_graph, num_graphs or None, cfg).  `seed` fixes the
    denominator graph (the model), `data_seed` (default = seed) the utterances: ranks of a
    data-parallel job share the graph and draw different utterances."""
    cfg = dict(CONFIGS[name])
    B, T, H, K, D = cfg["B"], cfg["T"], cfg["H"], cfg["K"], cfg["D"]
    ds = seed if data_seed is None else data_seed
    lengths = make_lengths(B, T, cfg["lengths"], seed=ds + 2)
    den = make_den_graph(H, K, D, seed=seed)
    num = make_num_graphs(lengths.tolist(), D, seed=ds + 100) if cfg["num"] else None
    x = make_input(B, T, D, seed=ds + 1, device=device)
    return dict(x=x, lengths=lengths, den_graph=den, num_graphs=num, cfg=cfg)

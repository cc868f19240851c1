"""How fast does the denominator's forward / backward filter forget where it started?  (fp64, CPU, numpy.)

TEST / ANALYSIS INFRASTRUCTURE ONLY (lives under oracle/: never imported by pychain_amd).

VERDICT r4 "Next round" 1(a): before any kernel cuts a sequence into time segments that start from a guessed state
vector, measure on the benchmark graphs how many burn-in frames n the alpha (beta) recursion of
chain-computation.cc:113-207 (:247-342) needs until a run started n frames before a splice point from the leaky
(all-ones) vector agrees with the run that started at frame 0 (frame T) to

    max_i |a_spec(i) - a_true(i)| / max_i a_true(i)  <=  tol          (both vectors normalised to sum 1)

for tol = 1e-7 (below fp32 rounding of the recursion itself), 1e-6, 1e-5; 20 seeds of network output per graph.
The recursions are the reference's equations (chain-computation.h:124-153) in float64:

    alpha'(t)   = alpha(t) + tot(t) * coef * leaky,                tot(t) = sum_i alpha(t, i)
    alpha(t+1,j) = sum_{k: dst=j} alpha'(t, src_k) p_k x(t, pdf_k) / tot(t)
    beta(t, i)  = sum_{k: src=i} p_k x(t, pdf_k) beta'(t+1, dst_k),   beta'(t) = beta(t) + coef * sum_i leaky_i beta(t,i)
                                                                       (any per-frame scale: only directions matter here)

    python oracle/forgetting.py            # writes profiles/r05_forgetting_table.md (+ .json)
"""
import json
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))

GRID = [2, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512]
TOLS = [1e-7, 1e-6, 1e-5]


def graph_arrays(g):
    ft = g.forward_transitions.numpy()
    return dict(src=ft[:, 0].astype(np.int64), dst=ft[:, 1].astype(np.int64), pdf=ft[:, 2].astype(np.int64),
                p=g.forward_transition_probs.numpy().astype(np.float64),
                leaky=g.leaky_probs.numpy().astype(np.float64), H=int(g.num_states))


def structured_graph(**kw):
    """pychain_amd.synthetic.make_structured_den_graph: a phone-LM-like denominator (arcs entering a state share its pdf,
    strong self-loops) - the filter mixes slower on it than on the random benchmark graph."""
    from pychain_amd import synthetic as syn
    return syn.make_structured_den_graph(**kw)


def frames(G, x, coef):
    """Per-frame arc weights w[t, k] = p_k * x(t, pdf_k) (float64)."""
    return G["p"][None, :] * x[:, G["pdf"]]


def alpha_run(G, w, a0, coef, t0, t1):
    """alpha from frame t0 (vector a0 = alpha(t0), un-dashed) to alpha(t1); returns normalised alpha(t1)."""
    a = a0.copy()
    for t in range(t0, t1):
        tot = a.sum()
        ad = a + tot * coef * G["leaky"]
        a = np.bincount(G["dst"], weights=ad[G["src"]] * w[t], minlength=G["H"]) / tot
    return a / a.sum()


def beta_run(G, w, b0, coef, t1, t0):
    """beta from frame t1 (vector b0 = beta'(t1)... any scale) backwards to beta(t0); normalised."""
    b = b0.copy()
    for t in range(t1 - 1, t0 - 1, -1):
        bd = b + coef * (G["leaky"] * b).sum()
        b = np.bincount(G["src"], weights=bd[G["dst"]] * w[t], minlength=G["H"])
        b = b / b.sum()
    return b


def burn_in(G, D, seeds, coef=1e-5, S=600, scale=2.0):
    """For each seed: the smallest grid n with error <= tol at the splice, alpha and beta."""
    from pychain_amd import synthetic as syn
    out = {"alpha": {tol: [] for tol in TOLS}, "beta": {tol: [] for tol in TOLS}, "err": {"alpha": [], "beta": []}}
    T = 2 * S
    ones = np.ones(G["H"])
    for seed in seeds:
        x = np.exp(np.clip(syn.normal(1000 + seed, T * D).reshape(T, D) * scale, -30, 30))
        w = frames(G, x, coef)
        a_true = alpha_run(G, w, G["leaky"], coef, 0, S)
        b_true = beta_run(G, w, ones, coef, T, S)
        ea, eb = [], []
        for n in GRID:
            a_s = alpha_run(G, w, G["leaky"], coef, S - n, S)
            b_s = beta_run(G, w, ones, coef, S + n, S)
            ea.append(float(np.abs(a_s - a_true).max() / a_true.max()))
            eb.append(float(np.abs(b_s - b_true).max() / b_true.max()))
        out["err"]["alpha"].append(ea)
        out["err"]["beta"].append(eb)
        for tol in TOLS:
            for name, e in (("alpha", ea), ("beta", eb)):
                ok = [n for n, v in zip(GRID, e) if v <= tol]
                # (errors fall monotonically; take the first n from which every larger n also holds)
                first = None
                for i, n in enumerate(GRID):
                    if all(v <= tol for v in e[i:]):
                        first = n
                        break
                out[name][tol].append(first if first is not None else -1)
    return out


def main():
    from pychain_amd import synthetic as syn
    n_seeds = int(os.environ.get("FORGET_SEEDS", "20"))
    cases = [("C2 graph (200 states / 2000 arcs, 1000 pdfs)", lambda: syn.make_den_graph(200, 2000, 1000), 1000, 60),
             ("C3 graph (3000 / 30000, 3456 pdfs)", lambda: syn.make_den_graph(3000, 30000, 3456), 3456, 600),
             ("C4 graph (3000 / 30000, 8408 pdfs)", lambda: syn.make_den_graph(3000, 30000, 8408), 8408, 600),
             ("structured phone-LM-like (3000 / 30000, 3456 pdfs, strong self-loops)", structured_graph, 3456, 600),
             ("structured, self-loop log-prob -0.05 (very sticky)", lambda: structured_graph(loop_lp=-0.05), 3456, 600)]
    res = {}
    lines = ["# Forgetting length of the denominator filter (fp64, `oracle/forgetting.py`)", "",
             "Smallest burn-in n (grid %s) after which a recursion started n frames before the splice from the leaky" % GRID,
             "(alpha) / all-ones (beta) vector agrees with the recursion from the sequence end to `tol` "
             "(max |diff| / max, normalised vectors);", "%d seeds of network output N(0,1)*2 per graph; leaky coefficient 1e-5. "
             "`max` over seeds is what a kernel has to provision." % n_seeds, "",
             "| graph | direction | tol 1e-7: median / max n | tol 1e-6 | tol 1e-5 | error at n=32 (max over seeds) | n=64 | n=128 |",
             "|---|---|---|---|---|---|---|---|"]
    for name, mk, D, S in cases:
        g = mk()
        G = graph_arrays(g)
        S = min(S, 600)
        if G["H"] <= 200:
            grid_ok = [n for n in GRID if n <= S]
        r = burn_in(G, D, range(n_seeds), S=max(S, 520))
        res[name] = {d: {str(t): r[d][t] for t in TOLS} for d in ("alpha", "beta")}
        res[name]["err"] = r["err"]
        for d in ("alpha", "beta"):
            e = np.asarray(r["err"][d])
            cells = []
            for tol in TOLS:
                v = np.asarray(r[d][tol])
                cells.append("%d / %s" % (int(np.median(v)), "not reached" if (v < 0).any() else int(v.max())))
            at = lambda n: "%.1e" % e[:, GRID.index(n)].max()
            lines.append("| %s | %s | %s | %s | %s | %s | %s | %s |" % (name, d, cells[0], cells[1], cells[2], at(32), at(64), at(128)))
        print(lines[-2]); print(lines[-1]); sys.stdout.flush()
    out = os.path.join(os.path.dirname(_HERE), "profiles")
    with open(os.path.join(out, "r05_forgetting_table.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(out, "r05_forgetting_table.json"), "w") as f:
        json.dump({"grid": GRID, "tols": TOLS, "seeds": n_seeds, "cases": res}, f)


if __name__ == "__main__":
    main()

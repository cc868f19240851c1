"""Time segments of the denominator recursions: step time of calls of the denominator alone for few sequences,
cut and not cut, and the worst mismatch of a speculated row against the burn-in on several graphs and input scales.
usage (GPU box): python tools/time_segments.py > gpurun_out/r05_time_segments.txt"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "tests")]
import torch
from helpers import rel_err
from pychain_amd import _lib, _plan, native, synthetic as syn
dev = torch.device("cuda:0")


def run(plan, x, L, **opts):
    ctx = [_lib.option(k, v) for k, v in opts.items()]
    for c in ctx: c.__enter__()
    try:
        objf, grad, bad, tot = native.den_forward_backward(plan, x, L, 1e-5, totals=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(6): native.den_forward_backward(plan, x, L, 1e-5, totals=True)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 6 * 1e3
    finally:
        for c in reversed(ctx): c.__exit__()
    return objf.clone(), grad.clone(), int(bad), tot.clone(), ms

print("== denominator alone, ms per call: not cut | 2 segments | 4 segments | automatic   (burn-in 192, the default; C3 graph ragged T<=1500, C4 T=2000)")
for name, B, T, mode in (("C3", 4, 1500, "ragged"), ("C3", 16, 1500, "ragged"), ("C3", 32, 1500, "ragged"), ("C4", 32, 2000, "equal"), ("C3", 48, 1500, "ragged"), ("C3", 64, 1500, "equal")):
    cfg = syn.CONFIGS[name]
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    plan = _plan.graph_plan(den, cfg["D"], dev)
    L = syn.make_lengths(B, T, mode, seed=3).to(dev)
    x = syn.make_input(B, T, cfg["D"], seed=9, device=dev)
    o0, g0, b0, t0, ms0 = run(plan, x, L, den_tseg=0)
    cells = ["%.3f" % ms0]
    for S in (2, 4, -1):
        o, g, b, t, ms = run(plan, x, L, den_tseg=S)
        cells.append("%.3f (S=%d, redone %d, worst %.1e, grad vs chain %.1e)" % (ms, int(t[6]), int(t[5]), float(t[7]), rel_err(g.cpu().numpy(), g0.cpu().numpy())))
    print("%s B=%d %s: " % (name, B, mode) + " | ".join(cells))
    native.release_workspaces(); torch.cuda.empty_cache()
print()
print("== worst mismatch of a speculated row (max |p - q| / max p; bound 4e-6), 4 segments, 3 draws of the network output, by burn-in")
for name, B, T in (("C3", 32, 1500), ("C4", 16, 2000)):
    cfg = syn.CONFIGS[name]
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    plan = _plan.graph_plan(den, cfg["D"], dev)
    L = torch.full((B,), T, device=dev)
    for scale in (1.0, 2.0, 3.0, 4.0):
        line = []
        for burn in (128, 192, 256, 384):
            worst, redo = 0.0, 0
            for seed in range(3):
                x = syn.make_input(B, T, cfg["D"], seed=40 + seed, scale=scale, device=dev)
                with _lib.option("den_tseg", 4), _lib.option("den_tburn", burn):
                    tot = native.den_forward_backward(plan, x, L, 1e-5, totals=True)[3]
                torch.cuda.synchronize()
                worst = max(worst, float(tot[7])); redo += int(tot[5])
            line.append("%d: %.1e%s" % (burn, worst, "" if not redo else " (%d rows missed)" % redo))
        print("%s graph, network output N(0,1) x %.0f: " % (name, scale) + " | ".join(line))

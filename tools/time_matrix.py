"""Time the fused step (or the denominator-only step for workloads without numerators) for a list of
(workload, options) cells in ONE process: tools/time_matrix.py "C3" "C3:den_dma=0" "C4" "C2:den_lazy=0" "C3@128" "C3:PLAN_LINEAR=1" ...
A cell is  WORKLOAD[@B][:opt=value[;opt=value...]] ; prints one line per cell (median of 5 groups of 6 steps) and, with
--parts, the recursion / occupancy launches in isolation."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch
from pychain_amd import ChainFunction, ChainGraphBatch, ChainLoss, _lib, _plan, native, synthetic as syn

dev = torch.device("cuda:0")
parts = "--parts" in sys.argv
cells = [a for a in sys.argv[1:] if not a.startswith("--")]
cache = {}


def workload(name, B, plan_env=()):
    key = (name, B, tuple(plan_env))
    for k in [k for k in os.environ if k.startswith("PYCHAIN_PLAN_") and k != "PYCHAIN_PLAN_CACHE_DIR"]:
        del os.environ[k]
    for k, v in plan_env:                              # plan compiler knobs are read when the plan is built
        os.environ["PYCHAIN_" + k] = v
    if key not in cache:
        cache.clear()
        torch.cuda.empty_cache()
        if B is None:
            w = syn.make_workload(name, device=dev)
        else:
            cfg = dict(syn.CONFIGS[name]); cfg["B"] = B
            L = syn.make_lengths(B, cfg["T"], cfg["lengths"], seed=2)
            w = dict(cfg=cfg, lengths=L, den_graph=syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0),
                     num_graphs=syn.make_num_graphs(L.tolist(), cfg["D"], seed=100) if cfg["num"] else None,
                     x=syn.make_input(B, cfg["T"], cfg["D"], seed=1, device=dev))
        w["Ld"] = w["lengths"].to(dev)
        cache[key] = w
    return cache[key]


def timed(fn, groups=5, per=6):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(groups)]
    for a, b in ev:
        a.record()
        for _ in range(per):
            fn()
        b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[groups // 2] / per


for cell in cells:
    head, _, optstr = cell.partition(":")
    name, _, bstr = head.partition("@")
    opts = dict(o.split("=", 1) for o in optstr.split(";") if o)
    plan_env = sorted((k, v) for k, v in opts.items() if k.startswith("PLAN_"))
    opts = {k: v for k, v in opts.items() if not k.startswith("PLAN_")}
    w = workload(name, int(bstr) if bstr else None, plan_env)
    cfg = w["cfg"]
    x = w["x"].requires_grad_(True)
    crit = ChainLoss(w["den_graph"], 1e-5, avg=False)
    gb = ChainGraphBatch(w["den_graph"], cfg["B"])

    def step():
        x.grad = None
        if w["num_graphs"] is not None:
            crit(x, w["Ld"], w["num_graphs"]).backward()
        else:
            ChainFunction.apply(x, w["Ld"], gb, 1e-5).backward()
    ctx = [_lib.option(k, v) for k, v in opts.items()]
    for c in ctx:
        c.__enter__()
    try:
        plan = _plan.graph_plan(w["den_graph"], cfg["D"], dev)
        names = _lib.den_kernel_names(plan.slot_rows, cfg["H"], cfg["D"], cfg["B"])
        ms = timed(step)
        frames = float(w["lengths"].sum())
        line = "%-28s %8.4f ms/step %7.2f M frames/s  [%s, %s]" % (cell, ms, frames / ms / 1e3, names[0], names[1])
        if parts:
            call = lambda: native.den_forward_backward(plan, w["x"].detach(), w["Ld"], 1e-5)
            for label, mask in (("rec", 1), ("occ", 2), ("den", 3)):
                # (the launches as the step above runs them: the fused loss never has its rows exp'd ahead - den_dma = 2)
                extra = [_lib.option("den_dma", 2)] if w["num_graphs"] is not None and "den_dma" not in opts else []
                for c in extra:
                    c.__enter__()
                try:
                    with _lib.option("den_phase_mask", mask):
                        line += "  %s %.4f" % (label, timed(call, 3, 3))
                finally:
                    for c in extra:
                        c.__exit__()
        bad = int(ChainFunction.last_bad_count.sum())
        print(line + ("  BAD=%d" % bad if bad else ""), flush=True)
    finally:
        for c in reversed(ctx):
            c.__exit__()

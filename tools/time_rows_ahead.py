"""Rows exp'd ahead of the recursions (den_exp_rows_kernel; the default of an uncut call of the denominator alone with at most 3/4 of the
CUs in recursion workgroups) against rows the recursions clamp / exp themselves (den_dma = 2), two-word and one-word state vectors:
the WHOLE call (both launches and the row kernel).  tools/time_rows_ahead.py [C2 | C3 B]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch
from pychain_amd import _lib, _plan, native, synthetic as syn
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
if name == "C3":
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    den = syn.make_den_graph(3000, 30000, 3456, seed=0)
    L = syn.make_lengths(B, 1500, "ragged", seed=2)
    x = syn.make_input(B, 1500, 3456, seed=1, device=dev)
    D = 3456; tag = "C3@B=%d" % B
elif name == "C4" and len(sys.argv) > 2:                 # C4's graph and rows (8408 pdfs) at another batch size, uncut
    B = int(sys.argv[2]); cfg = syn.CONFIGS["C4"]
    den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
    L = syn.make_lengths(B, 1000, "ragged", seed=2)
    x = syn.make_input(B, 1000, cfg["D"], seed=1, device=dev)
    D = cfg["D"]; tag = "C4@B=%d,T<=1000" % B
else:
    w = syn.make_workload(name, device=dev)
    den, L, x, D, tag = w["den_graph"], w["lengths"], w["x"], w["cfg"]["D"], name
plan = _plan.graph_plan(den, D, dev)
Ld = L.to(dev)
def med(f, n=9):
    f(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2]
call = lambda: native.den_forward_backward(plan, x, Ld, 1e-5)
for rep in range(2):
    for label, opts in (("default", {}), ("rows exp'd ahead (den_dma = 3)", {"den_dma": 3}), ("rows exp'd by the recursions (den_dma = 2)", {"den_dma": 2}),
                        ("one-word states, rows exp'd ahead", {"den_q": 1, "den_dma": 3}), ("one-word states, rows by the recursions", {"den_q": 1, "den_dma": 2})):
        ctx = [_lib.option(k, v) for k, v in opts.items()]
        for c in ctx: c.__enter__()
        try:
            names = _lib.den_kernel_names(plan.slot_rows, plan.num_states, D, x.size(0))[0]
            print("%-10s %-42s %.4f ms   (%s)" % (tag, label, med(call), names))
        finally:
            for c in reversed(ctx): c.__exit__()

"""The call of the denominator alone at B = 64 (C3-equal: every sequence 1500 frames; C3: ragged): two time segments (the default) against the
uncut call with the streamed occupancy launch, rows exp'd by the recursions (the default since round 6) or ahead of them (den_dma = 3)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch
from pychain_amd import _lib, _plan, native, synthetic as syn
dev = torch.device("cuda:0")
den = syn.make_den_graph(3000, 30000, 3456, seed=0)
plan = _plan.graph_plan(den, 3456, dev)
x = syn.make_input(64, 1500, 3456, seed=1, device=dev)
def med(f, n=9):
    f(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2]
for mode in ("equal", "ragged"):
    L = syn.make_lengths(64, 1500, mode, seed=2)
    Ld = L.to(dev)
    call = lambda: native.den_forward_backward(plan, x, Ld, 1e-5)
    for rep in range(2):
        for label, opts in (("two time segments (default)", {}), ("uncut, rows by the recursions", {"den_tseg": 0}), ("uncut, rows exp'd ahead", {"den_tseg": 0, "den_dma": 3})):
            ctx = [_lib.option(k, v) for k, v in opts.items()]
            for c in ctx: c.__enter__()
            try:
                print("B=64 %-7s %-34s %.4f ms  %.2f M frames/s" % (mode, label, med(call), float(L.sum()) / med(call) / 1e3))
            finally:
                for c in reversed(ctx): c.__exit__()

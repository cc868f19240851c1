"""Crossing (DenArgs::xf) against the same call without it: per-frame gradient error by region, objf, bad count."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch
from pychain_amd import ChainFunction, ChainGraphBatch, _lib, synthetic as syn
D = 3456
den = syn.make_structured_den_graph()
lens = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "700,651".split(","))]
opts = dict(kv.split("=") for kv in sys.argv[2:])
B, T = len(lens), max(lens)
L = torch.tensor(lens)
x = syn.make_input(B, T, D, seed=90 + T, device="cuda:0")
def call(**o):
    ctx = [_lib.option(k, v) for k, v in o.items()]
    for c in ctx: c.__enter__()
    try:
        xx = x.clone().requires_grad_(True)
        out = ChainFunction.apply(xx, L, ChainGraphBatch(den, B), 1e-5)
        out.backward(); torch.cuda.synchronize()
    finally:
        for c in reversed(ctx): c.__exit__()
    return float(out.detach()), xx.grad, int(out.bad_count.sum()), out.totals_all.cpu()
o1, g1, b1, t1 = call(**dict(dict(den_cross=1), **opts))
o0, g0, b0, t0 = call(den_cross=0, **opts)
print("objf", o1, o0, "bad", b1, b0, "totals", t1.tolist()[5:], t0.tolist()[5:])
gm = float(g0.abs().max())
for b in range(B):
    err = (g1[b] - g0[b]).abs().amax(dim=1) / gm
    rs1, rs0 = g1[b].sum(dim=1), g0[b].sum(dim=1)
    Lb = lens[b]
    worst = torch.topk(err[:Lb], min(8, Lb))
    print("seq", b, "L", Lb, "max err", float(err[:Lb].max()), "at", worst.indices.tolist(), "vals", [round(float(v), 8) for v in worst.values])
    mid = Lb // 2
    for name, lo, hi in (("beta part", 0, mid - 24), ("band", mid - 24, mid + 24), ("alpha part", mid + 24, Lb)):
        if hi > lo: print("   %-10s frames [%d,%d): max err %.3g, row sums xf %.6f..%.6f ref %.6f" % (name, lo, hi, float(err[lo:hi].max()), float(rs1[lo:hi].min()), float(rs1[lo:hi].max()), float(rs0[lo:hi].mean())))
    if Lb < T: print("   padding max", float(g1[b, Lb:].abs().max()))

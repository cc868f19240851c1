"""A/B of the one-word state vectors (option den_q) under the library named by PYCHAIN_HIP_LIB: correctness against den_q = 0 on the
C3 batch, the recursion launch alone (rows clamped / exp'd by the recursions: den_dma = 2, the form of the fused step), the fused step."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch
from pychain_amd import ChainLoss, _lib, _plan, native, synthetic as syn
dev = torch.device("cuda:0")
w = syn.make_workload("C3", device=dev)
plan = _plan.graph_plan(w["den_graph"], w["cfg"]["D"], dev)
Ld = w["lengths"].to(dev)
L = _lib.lib()
tag = os.path.basename(os.environ.get("PYCHAIN_HIP_LIB", "shipped"))
def med(f, n=7):
    f(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); f(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[n // 2]
res = {}
with _lib.option("den_tseg", 0), _lib.option("den_dma", 2):
    for q in (0, 1):
        with _lib.option("den_q", q):
            o, g, bad = native.den_forward_backward(plan, w["x"], Ld, 1e-5)
            torch.cuda.synchronize()
            res[q] = (o.double().cpu(), g.double().cpu(), int(bad.item()))
    o0, g0, b0 = res[0]; o1, g1, b1 = res[1]
    print(tag, "objf rel %.2e grad max-rel %.2e bad %d %d" % (float((o1 - o0).abs().max() / o0.abs().max()), float((g1 - g0).abs().max() / g0.abs().max()), b0, b1))
    for rep in range(2):
        for q in (1, 0):
            with _lib.option("den_q", q):
                L.pychain_hip_set_den_phase_mask(1)
                t = med(lambda: native.den_forward_backward(plan, w["x"], Ld, 1e-5))
                L.pychain_hip_set_den_phase_mask(3)
                print(tag, "q=%d recursion launch (rows exp'd in the kernel) %.4f ms" % (q, t))
num = w["num_graphs"] if "num_graphs" in w else None
x = w["x"].clone().requires_grad_(True)
crit = ChainLoss(w["den_graph"], 1e-5, avg=False)
def step():
    x.grad = None
    crit(x, Ld, w["num_graphs"]).backward()
if num is not None:
    for rep in range(2):
        for q in (1, 0):
            with _lib.option("den_q", q):
                for _ in range(3): step()
                t = med(lambda: [step() for _ in range(8)], 5) / 8
                print(tag, "q=%d fused step %.4f ms" % (q, t))

"""Are the results of two builds of the library the same BITS?  tools/bits_vs_variant.py run <lib.so> <out.pt> writes objective and gradient
of the C3 denominator call (uncut, rows exp'd by the recursions) and of the fused C3 step under that library; `cmp a.pt b.pt` compares."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch
if sys.argv[1] == "cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        same = torch.equal(a[k], b[k])
        print(k, "bit-identical" if same else "DIFFER: max |d| %.3e" % float((a[k].double() - b[k].double()).abs().max()))
    sys.exit(0)
os.environ["PYCHAIN_HIP_LIB"] = sys.argv[2]
from pychain_amd import ChainLoss, _lib, _plan, native, synthetic as syn
dev = torch.device("cuda:0")
w = syn.make_workload("C3", device=dev)
plan = _plan.graph_plan(w["den_graph"], w["cfg"]["D"], dev)
Ld = w["lengths"].to(dev)
out = {}
with _lib.option("den_tseg", 0), _lib.option("den_dma", 2):
    o, g, bad = native.den_forward_backward(plan, w["x"], Ld, 1e-5)
    out["den_objf"], out["den_grad"] = o.cpu(), g.cpu()
x = w["x"].clone().requires_grad_(True)
loss = ChainLoss(w["den_graph"], 1e-5, avg=False)(x, Ld, w["num_graphs"])
loss.backward()
out["loss"], out["loss_grad"] = loss.detach().cpu(), x.grad.cpu()
torch.save(out, sys.argv[3])
print("wrote", sys.argv[3])

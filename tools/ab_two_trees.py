"""A/B of two checkouts of this repository on one box: python tools/ab_two_trees.py <tree> (run once per tree; each imports its own
pychain_amd and library).  C3 bench batch, the denominator recursion launch alone and the whole call, fp32 and bf16 rows.
profiles/r05_ab_no_const.txt was made with it (the other tree: a git worktree of the commit before)."""
import os, sys, time
tree = sys.argv[1]
sys.path[:0] = [tree]
os.environ["PYCHAIN_PLAN_CACHE_DIR"] = "/tmp/plans_" + os.path.basename(tree.rstrip("/"))
import torch
import pychain_amd
from pychain_amd import _lib, _plan, native, synthetic as syn
assert os.path.abspath(pychain_amd.__file__).startswith(os.path.abspath(tree)), pychain_amd.__file__
dev = torch.device("cuda:0")
w = syn.make_workload("C3", device=dev)
plan = _plan.graph_plan(w["den_graph"], 3456, dev)
Ld = w["lengths"].to(dev)
def t(x, mask, n=8):
    with _lib.option("den_phase_mask", mask), _lib.option("den_dma", 2), _lib.option("den_tseg", 0):
        call = lambda: native.den_forward_backward(plan, x, Ld, 1e-5)
        call(); torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in ev:
            a.record(); call(); b.record()
        torch.cuda.synchronize()
        return sorted(a.elapsed_time(b) for a, b in ev)[n // 2]
xf = w["x"]
xh = xf.to(torch.bfloat16)
print(tree, "abi", _lib.ABI_VERSION, "fp32 recursion %.4f  bf16 recursion %.4f  fp32 den %.4f  bf16 den %.4f" % (t(xf, 1), t(xh, 1), t(xf, 3), t(xh, 3)))
pad = torch.empty(37 * 1024 * 1024 + 4096, dtype=torch.uint8, device=dev)
xh2 = xf.to(torch.bfloat16)
print(tree, "bf16 recursion at another address %.4f" % t(xh2, 1))

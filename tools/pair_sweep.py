"""Fused ChainLoss step on the C3 graph around the batch size where two sequences per recursion workgroup start to pay
(option den_pair: automatic / never / always).  usage (GPU box): python tools/pair_sweep.py"""
import os, sys, time
sys.path[:0] = [os.environ.get("GRAFT_REPO_ROOT", "/root/repo")]
import torch
from pychain_amd import ChainLoss, _lib, native, synthetic as syn
dev = torch.device("cuda:0")
cfg = syn.CONFIGS["C3"]
den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
crit = ChainLoss(den, 1e-5, avg=False)
print("C3 graph, ragged T <= 1500, fused step ms: automatic | den_pair=0 | den_pair=1")
for B in (72, 80, 88, 96, 104, 112, 128, 144):
    L = syn.make_lengths(B, cfg["T"], "ragged", seed=2)
    num = syn.make_num_graphs(L.tolist(), cfg["D"], seed=100)
    x = syn.make_input(B, cfg["T"], cfg["D"], seed=1, device=dev).requires_grad_(True)
    Ld = L.to(dev)
    cells = []
    for pair in (None, "0", "1"):
        ctx = _lib.option("den_pair", pair) if pair is not None else None
        if ctx: ctx.__enter__()
        for _ in range(3):
            x.grad = None; crit(x, Ld, num).backward()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8):
            x.grad = None; crit(x, Ld, num).backward()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 8 * 1e3
        if ctx: ctx.__exit__()
        cells.append("%.2f (%.1f M/s)" % (ms, int(L.sum()) / ms / 1e3))
    print("B=%d: " % B + " | ".join(cells))
    del x, num
    native.release_workspaces(); torch.cuda.empty_cache()

"""Fused ChainLoss step on the C3 graph by batch size: automatic / not cut / 2 / 4 time segments (option den_tseg).
usage (GPU box): python tools/tseg_fused_sweep.py   - profiles/r05_time_segments_fused.txt"""
import os, sys, time
sys.path[:0] = [os.environ.get("GRAFT_REPO_ROOT", "/root/repo")]
import torch
from pychain_amd import ChainLoss, _lib, native, synthetic as syn
dev = torch.device("cuda:0")
cfg = syn.CONFIGS["C3"]
den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
crit = ChainLoss(den, 1e-5, avg=False)
print("C3 graph, ragged T <= 1500, fused step ms: automatic | not cut | 2 segments | 4 segments")
for B in (16, 24, 32, 40, 48, 56):
    L = syn.make_lengths(B, cfg["T"], "ragged", seed=2)
    num = syn.make_num_graphs(L.tolist(), cfg["D"], seed=100)
    x = syn.make_input(B, cfg["T"], cfg["D"], seed=1, device=dev).requires_grad_(True)
    Ld = L.to(dev)
    cells = []
    for ts in (None, "0", "2", "4"):
        ctx = _lib.option("den_tseg", ts) if ts is not None else None
        if ctx: ctx.__enter__()
        for _ in range(3):
            x.grad = None; crit(x, Ld, num).backward()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8):
            x.grad = None; crit(x, Ld, num).backward()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 8 * 1e3
        if ctx: ctx.__exit__()
        from pychain_amd import ChainFunction
        t8 = ChainFunction.last_totals_all.float().cpu().tolist()
        cells.append("%.2f (S=%d, redone %d)" % (ms, int(t8[6]), int(t8[5])))
    print("B=%d: " % B + " | ".join(cells))
    del x, num
    native.release_workspaces(); torch.cuda.empty_cache()

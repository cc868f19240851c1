"""The fused loss over large batches, as one call and in slices (DESIGN.md §4 "Slices"): ms per step on the C3 graph.
usage (GPU box): python tools/slices_sweep.py > gpurun_out/r05_slices.txt"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch
from pychain_amd import ChainLoss, _lib, native, synthetic as syn
dev = torch.device("cuda:0")
cfg = syn.CONFIGS["C3"]
den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
crit = ChainLoss(den, 1e-5, avg=False)
print("C3 graph, ragged T <= 1500, fused ChainLoss fwd+bwd, ms per step (M frames/s): one call | 2 slices | 3 slices | automatic")
for B in (128, 160, 192, 224, 256, 320, 384):
    L = syn.make_lengths(B, cfg["T"], "ragged", seed=2)
    num = syn.make_num_graphs(L.tolist(), cfg["D"], seed=100)
    x = syn.make_input(B, cfg["T"], cfg["D"], seed=1, device=dev).requires_grad_(True)
    Ld = L.to(dev)
    cells = []
    for sl in ("0", "2", "3", "-1"):
        with _lib.option("chain_slices", sl):
            for _ in range(2):
                x.grad = None
                crit(x, Ld, num).backward()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                x.grad = None
                crit(x, Ld, num).backward()
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        cells.append("%.2f (%.1f)" % (ms, int(L.sum()) / ms / 1e3))
    print("B=%d: " % B + " | ".join(cells))
    del x, num
    native.release_workspaces(); torch.cuda.empty_cache()

"""Wall time of the first fused ChainLoss steps on C3, one by one (synchronised), with a 3 s idle gap after step 8:
the GPU needs 4-5 steps (~15 ms) after an idle period to reach its steady clocks (4.2, 3.9, 3.6, 3.4 ... ms), which is what
the warm-up steps of bench.py are for.  Run on the GPU box: python tools/step_times.py"""
import sys, time, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pychain_amd import ChainFunction, ChainLoss, synthetic as syn
dev = torch.device("cuda:0")
t0 = time.time()
w = syn.make_workload("C3", device=dev, seed=0, data_seed=0)
torch.cuda.synchronize(); print("workload %.2f s" % (time.time() - t0))
x = w["x"].requires_grad_(True)
L = w["lengths"].to(dev)
loss_fn = ChainLoss(w["den_graph"], 1e-5, avg=False)
for i in range(14):
    torch.cuda.synchronize(); t = time.perf_counter()
    x.grad = None
    loss = loss_fn(x, L, w["num_graphs"]); loss.backward()
    torch.cuda.synchronize(); print("step %2d  %.3f ms" % (i, 1e3 * (time.perf_counter() - t)), flush=True)
    if i == 8: time.sleep(3.0)        # an idle gap: do the clocks fall?

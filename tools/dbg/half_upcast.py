import sys; sys.path.insert(0, ".")
import torch, time
from pychain_amd import ChainLoss, native, synthetic as syn
dev = torch.device("cuda:0")
w = syn.make_workload("C3", device=dev)
Ld = w["lengths"].to(dev)
crit = ChainLoss(w["den_graph"], 1e-5, avg=False)
for dt in (torch.float32, torch.bfloat16):
    for flag in (True, False):
        if dt == torch.float32 and not flag:
            continue
        native.HALF_ROWS = flag
        x = w["x"].detach().to(dt).clone().requires_grad_(True)
        def step():
            x.grad = None
            crit(x, Ld, w["num_graphs"]).backward()
        for _ in range(5): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): step()
        torch.cuda.synchronize()
        print(dt, "kernels read 2-byte rows" if flag else "host-side up-cast", "%.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
        native.release_workspaces()
native.HALF_ROWS = True

import sys; sys.path.insert(0, ".")
import torch, time
from pychain_amd import ChainLoss, native, synthetic as syn
dev = torch.device("cuda:0")
w = syn.make_workload("C3", device=dev)
Ld = w["lengths"].to(dev)
crit = ChainLoss(w["den_graph"], 1e-5, avg=False)
for dt in (torch.bfloat16, torch.float32, torch.bfloat16, torch.float16, torch.float32):
    x = w["x"].detach().to(dt).clone().requires_grad_(True)
    def step():
        x.grad = None
        crit(x, Ld, w["num_graphs"]).backward()
    for _ in range(3): step()
    torch.cuda.synchronize()
    evs = []
    t0 = time.perf_counter()
    host = []
    for i in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h0 = time.perf_counter()
        a.record(); step(); b.record()
        host.append((time.perf_counter() - h0) * 1e3)
        evs.append((a, b))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 8 * 1e3
    print(dt, "wall %.3f" % wall, "gpu", " ".join("%.2f" % a.elapsed_time(b) for a, b in evs))
    print("   host ms per step", " ".join("%.2f" % h for h in host))

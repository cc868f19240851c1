import sys; sys.path.insert(0, ".")
import torch, time
from pychain_amd import ChainLoss, native, synthetic as syn
dev = torch.device("cuda:0")
w = syn.make_workload("C3", device=dev)
Ld = w["lengths"].to(dev)
crit = ChainLoss(w["den_graph"], 1e-5, avg=False)
for dt in (torch.float32, torch.bfloat16):
    x = w["x"].detach().to(dt).clone().requires_grad_(True)
    def step():
        x.grad = None
        crit(x, Ld, w["num_graphs"]).backward()
    for _ in range(3): step()
    torch.cuda.synchronize()
    s0 = torch.cuda.memory_stats()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): step()
    b.record(); torch.cuda.synchronize()
    s1 = torch.cuda.memory_stats()
    print(dt, "ms/step %.3f" % (a.elapsed_time(b) / 5), "device allocs %d frees %d" % (s1["num_device_alloc"] - s0["num_device_alloc"], s1["num_device_free"] - s0["num_device_free"]),
          "reserved %.2f GB" % (s1["reserved_bytes.all.current"] / 1e9), "grad ptr %x" % x.grad.data_ptr(), "is_contig", x.grad.is_contiguous())
    # time backward alone
    loss = crit(x, Ld, w["num_graphs"]); torch.cuda.synchronize()
    x.grad = None
    a.record(); loss.backward(); b.record(); torch.cuda.synchronize()
    print("   backward alone %.3f ms" % a.elapsed_time(b))
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        step(); torch.cuda.synchronize()
    ev = [(e.key, e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total) for e in prof.key_averages()]
    ev = sorted(ev, key=lambda t: -t[1])[:8]
    print("   ", ev)

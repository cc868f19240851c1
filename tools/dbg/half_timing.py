import sys, os, time
sys.path.insert(0, ".")
import torch
from pychain_amd import ChainFunction, ChainGraphBatch, ChainLoss, _lib, _plan, native, synthetic as syn
dev = torch.device("cuda:0")
w = syn.make_workload("C3", device=dev)
Ld = w["lengths"].to(dev)
den, num = w["den_graph"], w["num_graphs"]
plan = _plan.graph_plan(den, 3456, dev)
gt = num.device_tensors(dev)
stream = torch.cuda.current_stream(dev)
def ev(fn, n=5):
    fn(); torch.cuda.synchronize()
    a = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in a:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in a)[n // 2]
for dt in (torch.float32, torch.bfloat16, torch.float16):
    x = w["x"].to(dt)
    xr = x.clone().requires_grad_(True)
    crit = ChainLoss(den, 1e-5, avg=False)
    def step():
        xr.grad = None
        crit(xr, Ld, num).backward()
    def fwd(g, mask):
        with _lib.option("den_phase_mask", mask):
            native.chain_loss_forward(plan, gt, 1, num.num_states, x, Ld, 1e-5, with_grad=g)
    print(dt, "step %.3f" % ev(step), "fused fwd+grad %.3f" % ev(lambda: fwd(True, 3)), "fused fwd only %.3f" % ev(lambda: fwd(False, 3)),
          "num only (mask0) grad %.3f nograd %.3f" % (ev(lambda: fwd(True, 0)), ev(lambda: fwd(False, 0))),
          "den rec only %.3f" % ev(lambda: fwd(True, 1)),
          "den alone %.3f" % ev(lambda: native.den_forward_backward(plan, x, Ld, 1e-5)))
    native.release_workspaces()

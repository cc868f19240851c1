import sys; sys.path.insert(0, ".")
import torch, time
from pychain_amd import ChainLoss, _lib, _plan, native, synthetic as syn
dev = torch.device("cuda:0")
w = syn.make_workload("C3", device=dev)
Ld = w["lengths"].to(dev)
den, num = w["den_graph"], w["num_graphs"]
crit = ChainLoss(den, 1e-5, avg=False)
plan = _plan.graph_plan(den, 3456, dev)
gt = num.device_tensors(dev)
def ev(fn, n=5):
    fn(); torch.cuda.synchronize()
    a = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for s, e in a:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    return sorted(s.elapsed_time(e) for s, e in a)[n // 2]
for dt in (torch.float32, torch.bfloat16):
    x = w["x"].detach().to(dt).clone().requires_grad_(True)
    def step():
        x.grad = None
        crit(x, Ld, num).backward()
    for _ in range(3): step()
    xd = x.detach()
    def fwd(g, mask):
        with _lib.option("den_phase_mask", mask):
            r = native.chain_loss_forward(plan, gt, 1, num.num_states, xd, Ld, 1e-5, with_grad=g)
        return r
    print(dt, "x ptr %x" % x.data_ptr(), "step %.3f" % ev(step), "fwd+grad %.3f" % ev(lambda: fwd(True, 3)), "fwd only %.3f" % ev(lambda: fwd(False, 3)),
          "num only grad %.3f nograd %.3f" % (ev(lambda: fwd(True, 0)), ev(lambda: fwd(False, 0))), "den rec %.3f" % ev(lambda: fwd(True, 1)))
    r = fwd(True, 3)
    print("   grad ptr %x  dws %x nws %x" % (r[3].grad.data_ptr(), r[3].den_ws.data_ptr(), r[3].num_ws.data_ptr()))

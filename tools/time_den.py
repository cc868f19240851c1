"""Time the two denominator launches in isolation (events on the launch stream)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch
from pychain_amd import _lib, _plan, native, synthetic as syn
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
dev = torch.device("cuda:0")
w = syn.make_workload(name, device=dev)
if os.environ.get("TIME_DEN_STRUCTURED"):        # the phone-LM-like graph of the same size (every arc carries the pdf of the state it enters)
    w["den_graph"] = syn.make_structured_den_graph(w["cfg"]["H"] // 2, (w["cfg"]["K"] // (w["cfg"]["H"] // 2) - 2) // 2, w["cfg"]["D"])
plan = _plan.graph_plan(w["den_graph"], w["cfg"]["D"], dev)
Ld = w["lengths"].to(dev)
L = _lib.lib()
if os.environ.get("TIME_DEN_LAZY") is not None:
    L.pychain_hip_set_den_lazy(int(os.environ["TIME_DEN_LAZY"]))
isexp = bool(int(os.environ.get("TIME_DEN_ISEXP", "0")))
xin = w["x"].clamp(-30, 30).exp() if isexp else w["x"]
call = lambda: native.den_forward_backward(plan, xin, Ld, 1e-5, input_is_exp=isexp)
call(); torch.cuda.synchronize()
modes = (("recursion", 1), ("gamma", 2), ("both", 3))
if os.environ.get("TIME_DEN_ONLY"):
    modes = tuple(m for m in modes if m[0] == os.environ["TIME_DEN_ONLY"])
for mname, mask in modes:
    L.pychain_hip_set_den_phase_mask(mask)
    call(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
    for a, b in ev:
        a.record(); call(); b.record()
    torch.cuda.synchronize()
    print(name, mname, "ms", sorted(a.elapsed_time(b) for a, b in ev)[2])
L.pychain_hip_set_den_phase_mask(3)

"""One workload, a few calls, for the profiler passes of round 6 (tools/profile_round6.sh):
    tools/time_call.py <label> [reps]
label: C3 (the fused bench step), C3-equal / C2 / C4 (calls of the denominator alone as bench.py's other_workloads make them),
C3@B=128 / C3@B=256 (the denominator alone at that batch: what `den_forward_backward` of those lines times), C3-num_compat (the fused
step with the numerator in the reference's own arithmetic)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch
import bench
from pychain_amd import ChainFunction, ChainGraphBatch, ChainLoss, _lib, _plan, native
label = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
name, B, kw, fused = "C3", None, {}, False
if label == "C3" or label == "C3-num_compat":
    fused = True
elif label == "C3-structured":                   # the phone-LM-like graph of the same size: the fused step
    kw, fused = dict(structured=True), True
elif label == "C3-structured-den":               # ... its denominator alone, every sequence 1500 frames
    kw = dict(structured=True, equal=True, den_only=True)
elif label == "C3-equal":
    kw = dict(equal=True, den_only=True)
elif label.startswith("C3@B="):
    B = int(label.split("=")[1])
elif label.startswith("C3-fused@B="):             # the fused step at that batch (bench.py: other_workloads C3@B=...)
    B, fused = int(label.split("=")[1]), True
elif label.startswith("C3-structured@B="):       # the structured graph's denominator alone at that batch
    B, kw = int(label.split("=")[1]), dict(structured=True)
else:
    name = label
w = bench._adhoc_workload(name, B, dev, **kw)
cfg = w["cfg"]
frames = int(w["lengths"].sum())
if fused:
    x = w["x"].requires_grad_(True)
    crit = ChainLoss(w["den_graph"], 1e-5, avg=False)
    def call():
        x.grad = None
        crit(x, w["lengths_dev"], w["num_graphs"]).backward()
else:
    plan = _plan.graph_plan(w["den_graph"], cfg["D"], dev)
    xd = w["x"].detach()
    call = lambda: native.den_forward_backward(plan, xd, w["lengths_dev"], 1e-5)
import contextlib
ctx = contextlib.ExitStack()
if label == "C3-num_compat":
    ctx.enter_context(_lib.option("num_compat", 1))
if not fused and cfg["num"]:
    # the denominator's call of a workload whose STEP is fused, in the form the step runs it (bench.py: other_workloads):
    # no rows exp'd ahead, cut into time segments as the fused call would be
    tsegs = int(_lib.lib().pychain_hip_den_time_segments(plan.stride, plan.slot_rows, plan.num_states, cfg["D"], cfg["B"], cfg["T"], 1))
    ctx.enter_context(_lib.option("den_dma", 2))
    ctx.enter_context(_lib.option("den_tseg", tsegs if tsegs > 1 else 0))
with ctx:
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        call()
    b.record()
    torch.cuda.synchronize()
print("%s frames %d ms_per_call %.4f" % (label, frames, a.elapsed_time(b) / reps))
if os.environ.get("TIME_CALL_PARTS") and not fused:
    for key, mask in (("recursion", 1), ("occupancy", 2)):
        with _lib.option("den_phase_mask", mask), ctx:
            call(); torch.cuda.synchronize()
            a.record()
            for _ in range(reps):
                call()
            b.record(); torch.cuda.synchronize()
            print("  %s ms %.4f" % (key, a.elapsed_time(b) / reps))

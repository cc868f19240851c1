"""Burn-in length of the time segments against splice misses and time (C3-equal: 64 x 1500 frames, two segments):
tools/tburn_try.py [structured]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch, bench
from pychain_amd import _lib, _plan, native
dev = torch.device("cuda:0")
w = bench._adhoc_workload("C3", None, dev, equal=True, den_only=True, structured=len(sys.argv) > 1)
plan = _plan.graph_plan(w["den_graph"], w["cfg"]["D"], dev)
xd = w["x"].detach()
for scale in (1.0, 3.0):
    xs = xd * scale
    for tb in (192, 128, 96, 64, 48, 32):
        with _lib.option("den_tseg", 2), _lib.option("den_tburn", tb):
            for _ in range(2): out = native.den_forward_backward(plan, xs, w["lengths_dev"], 1e-5, totals=True)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(8): out = native.den_forward_backward(plan, xs, w["lengths_dev"], 1e-5, totals=True)
            b.record(); torch.cuda.synchronize()
        print("scale", scale, "tburn", tb, "ms", round(a.elapsed_time(b) / 8, 4), "misses / segments / worst mismatch", [float(v) for v in out[3][5:8]], "bad", int(out[2]))

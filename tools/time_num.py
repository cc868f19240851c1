"""Time the numerator call in isolation."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch
from pychain_amd import _lib, native, synthetic as syn
dev = torch.device("cuda:0")
w = syn.make_workload("C3", device=dev)
gt = w["num_graphs"].device_tensors(dev)
Ld = w["lengths"].to(dev)
call = lambda: native.num_forward_backward(gt, 1, w["num_graphs"].num_states, w["x"], Ld, grad_mode=_lib.GRAD_LINEAR)
call(); torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
for a, b in ev:
    a.record(); call(); b.record()
torch.cuda.synchronize()
print("C3 numerator (num_fb + num_occ, linear gradient) ms", sorted(a.elapsed_time(b) for a, b in ev)[2])

"""Algorithmic bytes per call of a tools/time_call.py label: denominator 12 D + 8 (H + 1) per live
frame; a fused step adds the numerator's 8 U_n + 8 (H_n + 1)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch
from pychain_amd import synthetic as syn
label = sys.argv[1]
name, B, equal, fused = "C3", None, False, label in ("C3", "C3-num_compat", "C3-structured")
if label in ("C3-equal", "C3-structured-den"):           # (the structured graph has the sizes of C3's)
    equal = True
elif label.startswith("C3@B="):
    B = int(label.split("=")[1])
elif not fused:
    name = label
cfg = dict(syn.CONFIGS[name])
B = B or cfg["B"]
lengths = syn.make_lengths(B, cfg["T"], "equal" if equal else cfg["lengths"], seed=2)
frames = int(lengths.sum())
total = (12 * cfg["D"] + 8 * (cfg["H"] + 1)) * frames
if fused:
    ng = syn.make_num_graphs(lengths.tolist(), cfg["D"], seed=100)
    ft, fi = ng.forward_transitions, ng.forward_transition_indices
    for b, Lb in enumerate(lengths.tolist()):
        kused = int(fi[b, :, 1].max())
        U = int(torch.unique(ft[b, :kused, 2]).numel())
        Hn = int((fi[b, :, 1] > fi[b, :, 0]).sum())
        total += Lb * (8 * U + 8 * (Hn + 1))
print(total)

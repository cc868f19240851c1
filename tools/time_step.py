"""Time the fused ChainLoss step (fwd+bwd) on an ad-hoc shape: tools/time_step.py B T [lengths=ragged|equal] [H K D]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO]
import torch
from pychain_amd import ChainLoss, synthetic as syn
B, T = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "ragged"
H, K, D = (int(v) for v in sys.argv[4:7]) if len(sys.argv) > 6 else (3000, 30000, 3456)
dev = torch.device("cuda:0")
den = syn.make_den_graph(H, K, D, seed=0)
L = syn.make_lengths(B, T, mode, seed=2)
num = syn.make_num_graphs(L.tolist(), D, seed=100)
x = syn.make_input(B, T, D, seed=1, device=dev).requires_grad_(True)
Ld = L.to(dev)
crit = ChainLoss(den, 1e-5, avg=False)
def step():
    x.grad = None
    crit(x, Ld, num).backward()
for _ in range(4): step()
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
for a, b in ev:
    a.record()
    for _ in range(8): step()
    b.record()
torch.cuda.synchronize()
ms = sorted(a.elapsed_time(b) for a, b in ev)[2] / 8
print("B=%d T=%d %s H=%d K=%d D=%d: %.4f ms/step, %.2f M frames/s" % (B, T, mode, H, K, D, ms, float(L.sum()) / ms / 1e3))

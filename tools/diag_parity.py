"""Diagnostic: error of the HIP path and of the f32 oracle against the f64 oracle, den and num apart
(C3 graph, 4 ragged utterances up to T = 1500; output kept as profiles/r02_parity_c3.txt)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "oracle")]
import numpy as np, torch
import oracle as orc
from pychain_amd import ChainFunction, ChainGraphBatch, synthetic as syn

def hip(x, L, graphs, leaky=1e-5):
    xx = x.to("cuda:0").requires_grad_(True)
    o = ChainFunction.apply(xx, L, graphs, leaky); o.backward()
    return float(o), xx.grad.cpu().numpy()

def rel(a, b): return np.abs(a.astype(np.float64) - b).max() / np.abs(b).max()

cfg = syn.CONFIGS["C3"]
den = syn.make_den_graph(cfg["H"], cfg["K"], cfg["D"], seed=0)
L = torch.tensor([1500, 1201, 977, 902])
x = syn.make_input(4, 1500, cfg["D"], seed=1)
numg = syn.make_num_graphs(L.tolist(), cfg["D"], seed=100)
for name, graphs in (("den", ChainGraphBatch(den, 4)), ("num", numg)):
    o, g = hip(x, L, graphs)
    o32, g32 = orc.chain_function(x, L, graphs, flavour="f32")
    o64, g64 = orc.chain_function(x, L, graphs, flavour="f64")
    print(name, "objf hip %.4f f32 %.4f f64 %.4f" % (o, o32, o64))
    print(name, "grad err  hip-vs-f64 %.3e   f32oracle-vs-f64 %.3e   hip-vs-f32 %.3e" % (rel(g, g64), rel(g32, g64), rel(g, g32.astype(np.float64))))
    d = np.abs(g.astype(np.float64) - g64)
    b, t, n = np.unravel_index(d.argmax(), d.shape)
    print(name, "worst at", (b, t, n), "hip", g[b, t, n], "f64", g64[b, t, n], "f32", g32[b, t, n], "rowsum hip", g[b, t].sum())
    per_t = d.max(-1)
    print(name, "err by time (seq0) quartiles:", [float(per_t[0, i:i + 375].max()) for i in range(0, 1500, 375)])

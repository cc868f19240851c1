import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import numpy as np, torch
from helpers import long_case, rel_err
from pychain_amd import ChainFunction, ChainLoss, ChainLossFunction, _lib
import oracle as orc
c = long_case("fold_T751")
DEV = "cuda:0"
for overlap in (True, False):
    ChainLossFunction.overlap = overlap
    with _lib.option("num_compat", 1):
        xx = c["x"].to(DEV).requires_grad_(True)
        loss = ChainLoss(c["den"], c["leaky"], avg=True)(xx, c["lengths"], c["num"])
        torch.cuda.synchronize()
        print("overlap", overlap, "after forward bad", ChainFunction.last_bad_count.tolist(), float(loss))
        loss.backward()
        torch.cuda.synchronize()
        print("  after backward bad", ChainFunction.last_bad_count.tolist())
    g = xx.grad.cpu().numpy()
    rl, rg = orc.chain_loss(c["x"], c["lengths"], c["den"], c["num"], c["leaky"], avg=True, flavour="f32")
    print("  loss vs f32 oracle", float(loss), float(rl), "grad rel", rel_err(g, rg))
with _lib.option("num_compat", 1):
    xx = c["x"].to(DEV).requires_grad_(True)
    o = ChainFunction.apply(xx, c["lengths"], c["num"])
    torch.cuda.synchronize()
    print("num alone bad", ChainFunction.last_bad_count.tolist(), float(o))

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
cat > /tmp/c4dbg2.py <<'PY'
import sys, time, torch, faulthandler
faulthandler.dump_traceback_later(60, exit=True)
sys.path[:0] = ["."]
from pychain_amd import ChainFunction, ChainGraphBatch, _lib, _plan, native, synthetic as syn
dev = torch.device("cuda:0")
name = sys.argv[1]
w = syn.make_workload(name, device=dev)
cfg = w["cfg"]; Ld = w["lengths"].to(dev)
x = w["x"].requires_grad_(True)
gb = ChainGraphBatch(w["den_graph"], cfg["B"])
def step():
    x.grad = None
    ChainFunction.apply(x, Ld, gb, 1e-5).backward()
for n in (1, 2, 4, 8, 30):
    t0 = time.time()
    for i in range(n): step()
    torch.cuda.synchronize()
    print(name, n, "steps back to back: %.4f s" % (time.time() - t0), "bad", int(ChainFunction.last_bad_count.sum()), flush=True)
plan = _plan.graph_plan(w["den_graph"], cfg["D"], dev)
for mask in (1, 2, 3):
    with _lib.option("den_phase_mask", mask):
        t0 = time.time()
        for i in range(6): native.den_forward_backward(plan, w["x"].detach(), Ld, 1e-5)
        torch.cuda.synchronize()
        print("mask", mask, "%.4f s" % (time.time() - t0), flush=True)
PY
timeout 200 python /tmp/c4dbg2.py C4 > $O/r4m_c4.log 2>&1; echo "rc=$?" >> $O/r4m_c4.log
grep -v amdgpu.ids $O/r4m_c4.log | tail -30

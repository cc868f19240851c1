#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
cat > /tmp/c4dbg.py <<'PY'
import sys, time, torch
sys.path[:0] = ["."]
from pychain_amd import ChainFunction, ChainGraphBatch, _lib, synthetic as syn
dev = torch.device("cuda:0")
name = sys.argv[1]
w = syn.make_workload(name, device=dev)
cfg = w["cfg"]; Ld = w["lengths"].to(dev)
x = w["x"].requires_grad_(True)
gb = ChainGraphBatch(w["den_graph"], cfg["B"])
for i in range(4):
    t0 = time.time()
    x.grad = None
    y = ChainFunction.apply(x, Ld, gb, 1e-5)
    y.backward()
    torch.cuda.synchronize()
    print(name, "step", i, "%.3f s" % (time.time() - t0), "objf", float(y), "totals", ChainFunction.last_totals.tolist() if getattr(ChainFunction, "last_totals", None) is not None else None, flush=True)
PY
timeout 200 python /tmp/c4dbg.py C4 > $O/r4l_c4.log 2>&1; echo "rc=$?" >> $O/r4l_c4.log
grep -v amdgpu.ids $O/r4l_c4.log | tail -8
PYCHAIN_DEN_SEGMENTS=3 timeout 200 python /tmp/c4dbg.py C4 > $O/r4l_c4_gated.log 2>&1; echo "rc=$?" >> $O/r4l_c4_gated.log
grep -v amdgpu.ids $O/r4l_c4_gated.log | tail -6
echo "== variants"
for v in norowstore; do
  PYCHAIN_HIP_LIB=build/variants/lib_$v.so timeout 200 python tools/time_matrix.py --parts "C3" "C2" > $O/r4l_matrix_$v.log 2>&1
  echo "-- $v"; grep -v amdgpu.ids $O/r4l_matrix_$v.log
done
timeout 200 python tools/time_matrix.py --parts "C3" "C2" "C3@128" > $O/r4l_matrix.log 2>&1
grep -v amdgpu.ids $O/r4l_matrix.log

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 300 python tools/time_matrix.py --parts "C3" "C4" "C2" > $O/r4q_matrix.log 2>&1
grep -v amdgpu.ids $O/r4q_matrix.log
for f in build/variants/lib_*.so; do
  v=$(basename $f .so)
  PYCHAIN_HIP_LIB=$f timeout 200 python tools/time_matrix.py --parts "C3" "C4" > $O/r4q_matrix_$v.log 2>&1
  echo "-- $v"; grep -v amdgpu.ids $O/r4q_matrix_$v.log
done
timeout 300 python tools/time_matrix.py --parts "C3" >> $O/r4q_matrix.log 2>&1
tail -1 $O/r4q_matrix.log

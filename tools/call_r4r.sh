#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > $O/r4r_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r4r_pytest.log
tail -3 $O/r4r_pytest.log
timeout 300 python tools/time_matrix.py --parts "C3" "C4" "C2" > $O/r4r_matrix.log 2>&1
grep -v amdgpu.ids $O/r4r_matrix.log
for f in build/variants/lib_nofinish.so; do
  v=$(basename $f .so)
  PYCHAIN_HIP_LIB=$f timeout 200 python tools/time_matrix.py --parts "C3" "C4" > $O/r4r_matrix_$v.log 2>&1
  echo "-- $v"; grep -v amdgpu.ids $O/r4r_matrix_$v.log
done

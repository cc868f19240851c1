#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_stream.py tests/test_gpu_robust.py tests/test_gpu_wide.py -m gpu -q -x > $O/r4n_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r4n_pytest.log
tail -3 $O/r4n_pytest.log
timeout 400 python tools/time_matrix.py --parts "C3" "C2" "C4" "C2@256" > $O/r4n_matrix.log 2>&1
grep -v amdgpu.ids $O/r4n_matrix.log
for v in storenoreread; do
  PYCHAIN_HIP_LIB=build/variants/lib_$v.so timeout 200 python tools/time_matrix.py --parts "C3" > $O/r4n_matrix_$v.log 2>&1
  echo "-- $v"; grep -v amdgpu.ids $O/r4n_matrix_$v.log
done
timeout 200 bash tools/phase_timers.sh run > $O/r4n_phase_timers.txt 2>&1
python tools/phase_table.py $O/r4n_phase_timers.txt 2>&1 | tail -45

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_general.py tests/test_gpu_wide.py tests/test_gpu_stream.py tests/test_gpu_ok.py -m gpu -q > $O/r4f_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r4f_pytest.log
tail -4 $O/r4f_pytest.log
python tools/time_matrix.py --parts "C2" "C2:den_segments=1" "C2@256" "C3" > $O/r4f_matrix.log 2>&1
grep -v amdgpu.ids $O/r4f_matrix.log

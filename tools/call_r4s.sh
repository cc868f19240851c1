#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 300 python -m pytest tests/test_gpu_wide.py -m gpu -q -x -k "exp_ahead" > $O/r4s_pytest0.log 2>&1; echo "pytest rc=$?" >> $O/r4s_pytest0.log
tail -15 $O/r4s_pytest0.log
timeout 300 python tools/time_matrix.py --parts "C3" "C4" "C2" "C3:den_dma=2" "C4:den_dma=2" > $O/r4s_matrix.log 2>&1
grep -v amdgpu.ids $O/r4s_matrix.log
timeout 900 python -m pytest tests -m gpu -q -x > $O/r4s_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r4s_pytest.log
tail -5 $O/r4s_pytest.log

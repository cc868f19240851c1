#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > $O/r4v_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r4v_pytest.log
tail -3 $O/r4v_pytest.log
timeout 400 python tools/time_matrix.py --parts "C3" "C4" "C2" "C4:den_dma=2" "C2:den_dma=2" "C3@128" "C3@256" "C2@256" "C3" > $O/r4v_matrix.log 2>&1
grep -v amdgpu.ids $O/r4v_matrix.log

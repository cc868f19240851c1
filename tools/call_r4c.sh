#!/bin/bash
# round 4, call c: GPU suite after the prune + small shape; matrix incl. C2 forms
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests -m gpu -q -x > $O/r4c_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r4c_pytest.log
tail -4 $O/r4c_pytest.log
python tools/time_matrix.py --parts "C3" "C2" "C2:den_segments=1" "C2:den_dma=0" "C2:den_lazy=0" "C4" "C3@128" "C3@256" "C2@256" > $O/r4c_matrix.log 2>&1
grep -v amdgpu.ids $O/r4c_matrix.log

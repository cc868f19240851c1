#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests -m gpu -q > $O/r4g_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r4g_pytest.log
tail -5 $O/r4g_pytest.log
cp $O/parity_measured.jsonl $O/r4g_parity_measured.jsonl 2>/dev/null
python tools/time_matrix.py --parts "C3" "C2" "C4" "C3@128" "C3@256" "C2@256" > $O/r4g_matrix.log 2>&1
grep -v amdgpu.ids $O/r4g_matrix.log

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
PYCHAIN_HIP_LIB=$PWD/tools/variants/phases.so TIME_DEN_ONLY=recursion PYCHAIN_DEN_SEGMENTS=1 python tools/time_den.py C2 2>&1 | grep -E "^lazy dir|recursion ms" | sort | uniq -c | sort -k3,3n -k5,5n > $O/r4e_phase_C2.txt
cat $O/r4e_phase_C2.txt | head -40
python -m pytest tests/test_gpu_wide.py -m gpu -q -k which_kernel 2>&1 | tail -2

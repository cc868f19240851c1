#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "LdsUtil" "LdsLatency" "SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  TIME_DEN_ONLY=recursion PYCHAIN_DEN_SEGMENTS=1 timeout 120 rocprofv3 --kernel-trace --pmc $ctrs -d $O/pmc_x$i -o p -- python $R/tools/time_den.py C3 > $O/pmc_x$i.log 2>&1
done
cd $R
python tools/pmc_report.py $O/pmc_x1/p_results.db $O/pmc_x2/p_results.db $O/pmc_x3/p_results.db $O/pmc_x4/p_results.db $O/pmc_x5/p_results.db $O/pmc_x6/p_results.db 2>&1 | grep -A24 "den_recursion_lazy" | head -40

#!/bin/bash
# SQ counters of the recursion launch on the structured graph: the one-gather form (PYCHAIN_DEN_SG unset) and the ordinary one (=0);
# and on the benchmark graph.  usage (GPU box): tools/pmc_sg.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export TIME_DEN_ONLY=recursion PYCHAIN_DEN_SEGMENTS=1 PYCHAIN_DEN_TSEG=0 PYCHAIN_DEN_DMA=2
for cfg in sg rnd; do
  if [ $cfg = sg ]; then export TIME_DEN_STRUCTURED=1; else unset TIME_DEN_STRUCTURED; fi
  i=0
  for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM"; do
    i=$((i+1))
    timeout 180 rocprofv3 --kernel-trace --pmc $ctrs -d $O/pmc_${tag}_${cfg}$i -o p -- python $R/tools/time_den.py C3 > $O/pmc_${tag}_${cfg}$i.log 2>&1
  done
  cd $R
  python tools/pmc_report.py $O/pmc_${tag}_${cfg}[1-5]/p_results.db 2>&1 | grep -A18 "den_recursion_lazy" > $O/${tag}_sq_counters_${cfg}.txt
  cat $O/${tag}_sq_counters_${cfg}.txt
  cd /tmp
done

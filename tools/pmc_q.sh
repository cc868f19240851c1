#!/bin/bash
# SQ counters of the C3 recursion launch with two-word (den_q = 0, the default) and one-word (den_q = 1) state vectors, rows clamped /
# exp'd by the recursions (den_dma = 2: the form of the fused step).  usage (GPU box): tools/pmc_q.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
export TIME_DEN_ONLY=recursion PYCHAIN_DEN_SEGMENTS=1 PYCHAIN_DEN_TSEG=0 PYCHAIN_DEN_DMA=2
for q in 0 1; do
  export PYCHAIN_DEN_Q=$q
  i=0
  for ctrs in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    timeout 180 rocprofv3 --kernel-trace --pmc $ctrs -d $O/pmc_${tag}_q$q$i -o p -- python $R/tools/time_den.py C3 > $O/pmc_${tag}_q$q$i.log 2>&1
  done
  cd $R
  python tools/pmc_report.py $O/pmc_${tag}_q$q[1-4]/p_results.db 2>&1 | grep -A14 "den_recursion_lazy" > $O/${tag}_sq_counters_q$q.txt
  echo "== den_q = $q"; cat $O/${tag}_sq_counters_q$q.txt
  rm -rf $O/pmc_${tag}_q$q[1-4] $O/pmc_${tag}_q$q[1-4].log
  cd /tmp
done

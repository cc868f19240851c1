#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
tag=r04
cd /tmp && export TMPDIR=/tmp
run_sq() {  # $1 = index, $2 = counters
  PYCHAIN_DEN_DMA=2 TIME_DEN_ONLY=recursion PYCHAIN_DEN_SEGMENTS=1 timeout 240 rocprofv3 --kernel-trace --pmc $2 -d $O/pmc_${tag}_sq$1 -o p -- python $R/tools/time_den.py C3 > $O/pmc_${tag}_sq$1.log 2>&1
}
run_sq 1 "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
for c in FETCH_SIZE WRITE_SIZE; do
  PYCHAIN_DEN_SEGMENTS=1 timeout 240 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${tag}_C4_$c -o p -- python $R/tools/time_den.py C4 > $O/pmc_${tag}_C4_$c.log 2>&1
done
run_sq 2 "SQ_INSTS_VALU SQ_INSTS_SALU"
run_sq 3 "SQ_WAVE_CYCLES SQ_WAIT_ANY"
cd $R
python tools/pmc_report.py $O/pmc_${tag}_sq1/p_results.db $O/pmc_${tag}_sq2/p_results.db $O/pmc_${tag}_sq3/p_results.db > $O/${tag}_sq_counters.txt 2>&1
python tools/traffic_json.py $O/pmc_${tag}_C4_FETCH_SIZE/p_results.db $O/pmc_${tag}_C4_WRITE_SIZE/p_results.db C4 64000 $O/${tag}_C4_hbm_traffic.json > /dev/null 2>&1
head -30 $O/${tag}_sq_counters.txt; grep -v _how $O/${tag}_C4_hbm_traffic.json | head -40

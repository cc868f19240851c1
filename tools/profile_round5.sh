#!/bin/bash
# usage (on the GPU box, via gpurun): tools/profile_round5.sh <tag>
# Round 5 (as round 4; the traffic files now hold den_exp_rows_kernel and the whole call's sum): kernel-trace stats of the default bench (C3, as shipped: streamed occupancy pass) and of the unsegmented
# schedule (one recursion launch, one occupancy launch per step: per-kernel averages comparable with
# roofline.ms_per_launch), the same for C4, and the HBM-traffic PMC passes (each counter its own rocprofv3 run with
# --pmc + --kernel-trace only, as the guide prescribes) for both workloads, plus the SQ counters of the recursion.
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
B="--steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads --no-fresh-num-graphs"
cd /tmp && export TMPDIR=/tmp
# C3 is the fused loss: its recursions exp their rows themselves (den_dma = 2 in the calls of the denominator alone that the
# counter passes time); C4 is a call of the denominator alone: rows exp'd ahead by den_exp_rows_kernel, as shipped
for wl in C3 C4; do
  if [ $wl = C3 ]; then export PYCHAIN_DEN_DMA=2; else unset PYCHAIN_DEN_DMA; fi
  rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_$wl -o bench -- python $R/bench.py --workload $wl $B > $O/prof_${tag}_$wl.log 2>&1
  PYCHAIN_DEN_SEGMENTS=1 rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_${wl}_unseg -o bench -- python $R/bench.py --workload $wl $B > $O/prof_${tag}_${wl}_unseg.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    PYCHAIN_DEN_SEGMENTS=1 rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${tag}_${wl}_$c -o p -- python $R/tools/time_den.py $wl > $O/pmc_${tag}_${wl}_$c.log 2>&1
  done
done
export PYCHAIN_DEN_DMA=2
i=0
for ctrs in "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  i=$((i+1))
  TIME_DEN_ONLY=recursion PYCHAIN_DEN_SEGMENTS=1 rocprofv3 --kernel-trace --pmc $ctrs -d $O/pmc_${tag}_sq$i -o p -- python $R/tools/time_den.py C3 > $O/pmc_${tag}_sq$i.log 2>&1
done
unset PYCHAIN_DEN_DMA
cd $R
for wl in C3 C4; do
  python tools/rocpd_stats.py $O/prof_${tag}_$wl/bench_results.db $O/${tag}_${wl}_kernel_stats.md > /dev/null
  python tools/rocpd_stats.py $O/prof_${tag}_${wl}_unseg/bench_results.db $O/${tag}_${wl}_unsegmented_kernel_stats.md > /dev/null
  frames=$(python -c "from pychain_amd import synthetic as s; import sys; print(int(s.make_lengths(s.CONFIGS['$wl']['B'], s.CONFIGS['$wl']['T'], s.CONFIGS['$wl']['lengths'], seed=2).sum()))")
  python tools/traffic_json.py $O/pmc_${tag}_${wl}_FETCH_SIZE/p_results.db $O/pmc_${tag}_${wl}_WRITE_SIZE/p_results.db $wl $frames $O/${tag}_${wl}_hbm_traffic.json > /dev/null 2>&1
  tail -1 $O/prof_${tag}_$wl.log | cut -c1-400
done
python tools/pmc_report.py $O/pmc_${tag}_sq1/p_results.db $O/pmc_${tag}_sq2/p_results.db $O/pmc_${tag}_sq3/p_results.db > $O/${tag}_sq_counters.txt 2>&1
cat $O/${tag}_C3_kernel_stats.md | head -14; cat $O/${tag}_C4_kernel_stats.md | head -8; cat $O/${tag}_C3_hbm_traffic.json $O/${tag}_C4_hbm_traffic.json | grep -v _how; head -20 $O/${tag}_sq_counters.txt

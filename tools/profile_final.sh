#!/bin/bash
# usage (on the GPU box, via gpurun): tools/profile_final.sh <tag>
# The last pass of a round: kernel-trace stats of the bench as shipped and unsegmented (profiles/<tag>_C3_*kernel_stats.md) and the
# HBM counter traffic of the pdf-by-state graph's step and denominator call (separate --pmc passes, as the guide prescribes; the
# counter passes run the unsegmented schedule: tools/profile_round6.sh says why).
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
REPS=5
cd /tmp && export TMPDIR=/tmp
B="--steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads --no-fresh-num-graphs"
rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_C3 -o bench -- python $R/bench.py --workload C3 $B > $O/prof_${tag}_C3.log 2>&1
PYCHAIN_DEN_SEGMENTS=1 rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_C3_unseg -o bench -- python $R/bench.py --workload C3 $B > $O/prof_${tag}_C3_unseg.log 2>&1
export PYCHAIN_DEN_SEGMENTS=1
for lb in C3-structured C3-structured-den; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${tag}_${lb}_$c -o p -- python $R/tools/time_call.py $lb $REPS > $O/pmc_${tag}_${lb}_$c.log 2>&1
  done
done
unset PYCHAIN_DEN_SEGMENTS
cd $R
python tools/rocpd_stats.py $O/prof_${tag}_C3/bench_results.db $O/${tag}_C3_kernel_stats.md > /dev/null
python tools/rocpd_stats.py $O/prof_${tag}_C3_unseg/bench_results.db $O/${tag}_C3_unsegmented_kernel_stats.md > /dev/null
for lb in C3-structured C3-structured-den; do
  frames=$(grep ' frames ' $O/pmc_${tag}_${lb}_FETCH_SIZE.log | tail -1 | awk '{print $3}')
  python tools/step_traffic_json.py $O/pmc_${tag}_${lb}_FETCH_SIZE/p_results.db $O/pmc_${tag}_${lb}_WRITE_SIZE/p_results.db $lb $frames $((REPS+2)) \
     $(python tools/algorithmic_bytes.py $lb) $O/${tag}_${lb}_step_hbm_traffic.json | grep -v '"_how"' | head -40
done
head -12 $O/${tag}_C3_kernel_stats.md

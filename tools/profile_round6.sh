#!/bin/bash
# usage (on the GPU box, via gpurun): tools/profile_round6.sh <tag> [labels...]
# Round 6: HBM counter traffic of EVERY kernel of the fused C3 step - the numerator's kernels included -
# and of the calls behind the other_workloads lines that used to say "traffic": null (C3-equal, C2, C3@B=128, C3@B=256), plus
# the kernel-trace stats of the bench as shipped and unsegmented.  Each counter is its own rocprofv3 run with --pmc +
# --kernel-trace only, as the guide prescribes.
tag=$1; shift
labels=${@:-"C3 C3-equal C2 C3@B=128 C3@B=256"}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
REPS=5
cd /tmp && export TMPDIR=/tmp
# (counter collection serialises the dispatches of a process: a schedule whose kernels wait for one another ACROSS streams - the
# streamed occupancy launch, den_finish_kernel polling its counter - would sit out its 20 s time-outs (the first run of this script
# did: same bytes, `bad` counted).  The counter passes therefore run the unsegmented schedule: the same kernels' bytes, one after the other.)
export PYCHAIN_DEN_SEGMENTS=1
for lb in $labels; do
  f=$(echo $lb | tr '@=' '__')
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${tag}_${f}_$c -o p -- python $R/tools/time_call.py $lb $REPS > $O/pmc_${tag}_${f}_$c.log 2>&1
  done
done
unset PYCHAIN_DEN_SEGMENTS
B="--steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads --no-fresh-num-graphs"
rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_C3 -o bench -- python $R/bench.py --workload C3 $B > $O/prof_${tag}_C3.log 2>&1
PYCHAIN_DEN_SEGMENTS=1 rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_C3_unseg -o bench -- python $R/bench.py --workload C3 $B > $O/prof_${tag}_C3_unseg.log 2>&1
cd $R
python tools/rocpd_stats.py $O/prof_${tag}_C3/bench_results.db $O/${tag}_C3_kernel_stats.md > /dev/null
python tools/rocpd_stats.py $O/prof_${tag}_C3_unseg/bench_results.db $O/${tag}_C3_unsegmented_kernel_stats.md > /dev/null
for lb in $labels; do
  f=$(echo $lb | tr '@=' '__')
  frames=$(grep ' frames ' $O/pmc_${tag}_${f}_FETCH_SIZE.log | tail -1 | awk '{print $3}')
  python tools/step_traffic_json.py $O/pmc_${tag}_${f}_FETCH_SIZE/p_results.db $O/pmc_${tag}_${f}_WRITE_SIZE/p_results.db $lb $frames $((REPS+2)) \
     $(python tools/algorithmic_bytes.py $lb) $O/${tag}_${f}_step_hbm_traffic.json | grep -v '"_how"' | head -60
done
head -14 $O/${tag}_C3_kernel_stats.md

#!/bin/bash
# usage (on the GPU box, via gpurun): tools/profile_final.sh <tag>
# Final state of round 5: kernel-trace stats of the default bench (C3) and of C4 as shipped - C4's call of the denominator
# alone is now cut into four time segments (no den_exp_rows_kernel, a splice check, a fallback launch that
# leaves at once) - and the two HBM-traffic PMC passes of that call (each counter its own rocprofv3 run with --pmc +
# --kernel-trace only, as the guide prescribes).
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
B="--steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads --no-fresh-num-graphs"
cd /tmp && export TMPDIR=/tmp
for wl in C3 C4; do
  rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_$wl -o bench -- python $R/bench.py --workload $wl $B > $O/prof_${tag}_$wl.log 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  TIME_DEN_ONLY=both rocprofv3 --kernel-trace --pmc $c -d $O/pmc_${tag}_C4_$c -o p -- python $R/tools/time_den.py C4 > $O/pmc_${tag}_C4_$c.log 2>&1
done
cd $R
for wl in C3 C4; do
  python tools/rocpd_stats.py $O/prof_${tag}_$wl/bench_results.db $O/${tag}_${wl}_kernel_stats.md > /dev/null
  tail -1 $O/prof_${tag}_$wl.log | cut -c1-300
done
python tools/traffic_json.py $O/pmc_${tag}_C4_FETCH_SIZE/p_results.db $O/pmc_${tag}_C4_WRITE_SIZE/p_results.db C4 64000 $O/${tag}_C4_hbm_traffic.json > /dev/null 2>&1
head -12 $O/${tag}_C3_kernel_stats.md; head -10 $O/${tag}_C4_kernel_stats.md; grep -v _how $O/${tag}_C4_hbm_traffic.json

#!/bin/bash
# usage (on the GPU box, via gpurun): tools/profile_round.sh <tag>
# kernel-trace stats of the default bench (as shipped) and of the unsegmented schedule, plus the two
# HBM-traffic PMC passes (each its own rocprofv3 run, as the guide prescribes).
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag} -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_${tag}.log 2>&1
PYCHAIN_DEN_SEGMENTS=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag}_unseg -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_${tag}_unseg.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  PYCHAIN_DEN_SEGMENTS=1 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_${tag}_$c -o p -- python $R/tools/time_den.py C3 > $R/gpurun_out/pmc_${tag}_$c.log 2>&1
done
cd $R
python tools/rocpd_stats.py gpurun_out/prof_${tag}/bench_results.db gpurun_out/${tag}_kernel_stats.md
python tools/rocpd_stats.py gpurun_out/prof_${tag}_unseg/bench_results.db gpurun_out/${tag}_unseg_kernel_stats.md
python tools/pmc_report.py gpurun_out/pmc_${tag}_FETCH_SIZE/p_results.db gpurun_out/pmc_${tag}_WRITE_SIZE/p_results.db > gpurun_out/${tag}_pmc.txt 2>&1
python tools/traffic_json.py gpurun_out/pmc_${tag}_FETCH_SIZE/p_results.db gpurun_out/pmc_${tag}_WRITE_SIZE/p_results.db C3 76684 gpurun_out/${tag}_hbm_traffic.json > /dev/null 2>&1
tail -2 gpurun_out/prof_${tag}.log; cat gpurun_out/${tag}_pmc.txt; ls gpurun_out/prof_${tag} gpurun_out/pmc_${tag}_FETCH_SIZE

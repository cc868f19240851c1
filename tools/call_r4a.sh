#!/bin/bash
# round 4, call a: GPU suite on the G6 / ADVICE state, same-box baselines, launch list of a step
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests -m gpu -x -q > $O/r4a_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r4a_pytest.log
tail -3 $O/r4a_pytest.log
cp $O/parity_measured.jsonl $O/r4a_parity_measured.jsonl 2>/dev/null
python tools/time_matrix.py --parts "C3" "C3:PLAN_LINEAR=1" "C2" "C4" "C3@128" "C3@256" > $O/r4a_matrix.log 2>&1
cat $O/r4a_matrix.log | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/$O/prof_r4a_C3 -o bench -- python $R/bench.py --workload C3 --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads --no-fresh-num-graphs > $R/$O/prof_r4a_C3.log 2>&1
cd $R
python tools/rocpd_stats.py $O/prof_r4a_C3/bench_results.db $O/r4a_C3_kernel_stats.md > /dev/null 2>&1
head -30 $O/r4a_C3_kernel_stats.md
python tools/timeline.py $O/prof_r4a_C3/bench_results.db > $O/r4a_C3_timeline.txt 2>&1
rm -rf $O/prof_r4a_C3

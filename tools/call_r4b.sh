#!/bin/bash
# round 4, call b: whole GPU suite (G6 bounds, pickled batches, step totals), bench line, step timeline
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests -m gpu -q > $O/r4b_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r4b_pytest.log
tail -4 $O/r4b_pytest.log
cp $O/parity_measured.jsonl $O/r4b_parity_measured.jsonl 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads > $O/r4b_bench.log 2>&1; tail -1 $O/r4b_bench.log | cut -c1-600
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/$O/prof_r4b_C3 -o bench -- python $R/bench.py --workload C3 --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads --no-fresh-num-graphs --no-rooflines > $R/$O/prof_r4b_C3.log 2>&1
cd $R
python tools/rocpd_stats.py $O/prof_r4b_C3/bench_results.db $O/r4b_C3_kernel_stats.md > /dev/null 2>&1
python tools/timeline.py $O/prof_r4b_C3/bench_results.db > $O/r4b_C3_timeline.txt 2>&1
cat $O/r4b_C3_timeline.txt | head -30
rm -rf $O/prof_r4b_C3

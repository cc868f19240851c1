#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 270 python bench.py --no-other-workloads > gpurun_out/r04_bench_line_final_commit.json 2> gpurun_out/r04_bench_final_err.log; echo "rc=$?"
tail -c 1500 gpurun_out/r04_bench_line_final_commit.json

#!/bin/bash
# What the driver runs at round end, in one gpurun call: build() + smoke(), the whole GPU suite, the default bench line (gpurun_out/final_smoke.txt,
# full_gpu_suite.txt, bench_final.json).  usage: gpurun --timeout 3000 -- bash tools/final_check.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/final_smoke.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/full_gpu_suite.txt
cat gpurun_out/final_smoke.txt gpurun_out/full_gpu_suite.txt
timeout 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; python -c "
import json; d=json.load(open('gpurun_out/bench_final.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['ms_per_launch'])"

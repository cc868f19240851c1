cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_round6.sh r06c C3 > gpurun_out/profile_r06c.log 2>&1
rm -rf gpurun_out/pmc_r06c_* gpurun_out/prof_r06c_*
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/full_gpu_suite.txt
cat gpurun_out/full_gpu_suite.txt
timeout 600 python bench.py > gpurun_out/bench_r06c.json 2> gpurun_out/bench_r06c.err; tail -c 300 gpurun_out/bench_r06c.json

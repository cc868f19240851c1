#!/bin/bash
# round 4, call d: small shape with the two-ahead ring; affected test files; matrix; bench line with numerator rooflines
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_wide.py tests/test_gpu_stream.py tests/test_gpu_ok.py tests/test_gpu_configs.py tests/test_gpu_random.py tests/test_gpu_robust.py -m gpu -q > $O/r4d_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r4d_pytest.log
tail -4 $O/r4d_pytest.log
python tools/time_matrix.py --parts "C2" "C2:den_segments=1" "C2:den_dma=0" "C2@256" "C3" > $O/r4d_matrix.log 2>&1
grep -v amdgpu.ids $O/r4d_matrix.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads > $O/r4d_bench.log 2>&1; tail -1 $O/r4d_bench.log | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print(j['value'], j['ms_per_step'], json.dumps(j['roofline']['other_kernels'], indent=0)[:1500])"

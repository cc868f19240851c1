cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
python tools/q_variants.py 2>&1 | grep -v amdgpu.ids
PYCHAIN_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/prev.so python tools/q_variants.py 2>&1 | grep -v amdgpu.ids
python tools/q_variants.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/valu_trim2.txt 2>&1
cat gpurun_out/valu_trim2.txt
timeout 1200 python -m pytest tests/test_gpu_q.py tests/test_gpu_parity.py tests/test_gpu_ok.py tests/test_gpu_tseg.py tests/test_gpu_random.py -x -q -m gpu 2>&1 | tail -5

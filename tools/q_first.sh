cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in short0 shortA shortB short0; do PYCHAIN_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/$v.so python tools/q_variants.py 2>&1 | grep -v amdgpu.ids | grep "q=0\|objf"; done > gpurun_out/short_waves.txt 2>&1
cat gpurun_out/short_waves.txt

#!/bin/bash
# usage (GPU box): tools/ab_variants.sh <out-name> <variant...>: tools/q_variants.py (den_q = 0 lines) under tools/variants/<variant>.so,
# the shipped library ("shipped") before, between and after
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
out=$1; shift
for v in shipped "$@" shipped; do if [ $v = shipped ]; then python tools/q_variants.py; else PYCHAIN_HIP_LIB=$GRAFT_REPO_ROOT/tools/variants/$v.so python tools/q_variants.py; fi 2>&1 | grep -v amdgpu.ids | grep "q=0\|objf"; done > gpurun_out/$out.txt 2>&1
cat gpurun_out/$out.txt

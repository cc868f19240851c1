#!/bin/bash
# usage: tools/pmc_run.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...]   (run on the GPU box)
# Each pass is its own rocprofv3 run with --pmc only (+ kernel trace), as the guide prescribes.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_p$i -o p -- python $GRAFT_REPO_ROOT/tools/time_den.py C3 > $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_p$i.log 2>&1
done
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_p*/

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 300 python -m pytest tests/test_gpu_wide.py -m gpu -q -x -k "exp_ahead" > $O/r4u_pytest0.log 2>&1; echo "pytest rc=$?" >> $O/r4u_pytest0.log
tail -3 $O/r4u_pytest0.log
timeout 300 python tools/time_matrix.py --parts "C3" "C4" "C2" "C3:den_dma=2" "C4:den_dma=2" "C2:den_dma=2" "C3" > $O/r4u_matrix.log 2>&1
grep -v amdgpu.ids $O/r4u_matrix.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cell in "C3" "C4"; do
  tag=$(echo $cell | tr '@' '_')
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_r4u_$tag -o m -- python $R/tools/time_matrix.py "$cell" > $R/$O/prof_r4u_$tag.log 2>&1
  python $R/tools/timeline.py $R/$O/prof_r4u_$tag/m_results.db > $R/$O/r4u_${tag}_timeline.txt 2>&1
  rm -rf $R/$O/prof_r4u_$tag
  echo "== $cell"; head -16 $R/$O/r4u_${tag}_timeline.txt
done

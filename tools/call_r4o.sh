#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > $O/r4o_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r4o_pytest.log
tail -3 $O/r4o_pytest.log
timeout 500 python tools/time_matrix.py --parts "C3" "C2" "C4" "C2@256" "C3@128" "C3@256" > $O/r4o_matrix.log 2>&1
grep -v amdgpu.ids $O/r4o_matrix.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cell in "C3" "C2"; do
  tag=$(echo $cell | tr '@' '_')
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_r4o_$tag -o m -- python $R/tools/time_matrix.py "$cell" > $R/$O/prof_r4o_$tag.log 2>&1
  python $R/tools/timeline.py $R/$O/prof_r4o_$tag/m_results.db > $R/$O/r4o_${tag}_timeline.txt 2>&1
  rm -rf $R/$O/prof_r4o_$tag
  echo "== $cell"; head -16 $R/$O/r4o_${tag}_timeline.txt
done

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > $O/r4k_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r4k_pytest.log
tail -3 $O/r4k_pytest.log
timeout 300 python tools/time_matrix.py --parts "C3" "C2" "C4" "C3@128" > $O/r4k_matrix.log 2>&1
grep -v amdgpu.ids $O/r4k_matrix.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cell in "C3" "C2"; do
  tag=$(echo $cell | tr '@' '_')
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_r4k_$tag -o m -- python $R/tools/time_matrix.py "$cell" > $R/$O/prof_r4k_$tag.log 2>&1
  python $R/tools/timeline.py $R/$O/prof_r4k_$tag/m_results.db > $R/$O/r4k_${tag}_timeline.txt 2>&1
  rm -rf $R/$O/prof_r4k_$tag
  echo "== $cell"; head -24 $R/$O/r4k_${tag}_timeline.txt
done

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py tests/test_gpu_general.py -m gpu -q > $O/r4h_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r4h_pytest.log
tail -3 $O/r4h_pytest.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cell in "C3@256" "C3@128"; do
  tag=$(echo $cell | tr '@' '_')
  rocprofv3 --kernel-trace --stats -d $R/$O/prof_r4h_$tag -o m -- python $R/tools/time_matrix.py "$cell" > $R/$O/prof_r4h_$tag.log 2>&1
  python $R/tools/timeline.py $R/$O/prof_r4h_$tag/m_results.db > $R/$O/r4h_${tag}_timeline.txt 2>&1
  python $R/tools/rocpd_stats.py $R/$O/prof_r4h_$tag/m_results.db $R/$O/r4h_${tag}_kernel_stats.md > /dev/null 2>&1
  rm -rf $R/$O/prof_r4h_$tag
  echo "== $cell"; head -24 $R/$O/r4h_${tag}_timeline.txt
done

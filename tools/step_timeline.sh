# kernel timeline of one step (rocprofv3 --kernel-trace + tools/timeline.py): tools/tl_structured.sh <time_call label> ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in "$@"; do
rocprofv3 --kernel-trace -d $R/gpurun_out/tl_$w -o p -- python $R/tools/time_call.py $w 6 > /dev/null 2>&1
echo "== $w"; python $R/tools/timeline.py $R/gpurun_out/tl_$w/p_results.db 2>&1 | tail -16
done

mkdir -p gpurun_out/r03s
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03s/kt -o ss -- python3 $GRAFT_REPO_ROOT/tools/steady_state.py 1200 > $GRAFT_REPO_ROOT/gpurun_out/r03s/ss.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/r03s/ss.log | cut -c1-150
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r03s/kt -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -16 "$f" | cut -c1-200
find $GRAFT_REPO_ROOT/gpurun_out/r03s/kt -name "*kernel_trace.csv" -size +30M -delete

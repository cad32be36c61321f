#!/bin/bash
# rocprofv3 kernel trace of the steady-state pipeline (tools/steady_state.py); writes under gpurun_out/r05ss
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/gpurun_out/r05ss
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r05ss/kt -o ss -- python3 $ROOT/tools/steady_state.py 1200 > $ROOT/gpurun_out/r05ss/ss.log 2>&1
tail -1 $ROOT/gpurun_out/r05ss/ss.log | cut -c1-150
f=$(find $ROOT/gpurun_out/r05ss/kt -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -16 "$f" | cut -c1-200
find $ROOT/gpurun_out/r05ss/kt -name "*kernel_trace.csv" -size +30M -delete

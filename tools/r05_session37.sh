#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05s37; mkdir -p $OUT
{
echo "--- steady state, the forms in memory"; python3 tools/steady_state2.py 1 4000
echo "--- steady state, the matrix in registers"; FRANK_AMD_K2_RR=1 python3 tools/steady_state2.py 1 4000
FRANK_AMD_K2_RR=1 FRANK_AMD_FIT_SLOTS=360 python3 tools/steady_state2.py 1 4000
FRANK_AMD_K2_RR=1 FRANK_AMD_FIT_SLOTS=480 FRANK_AMD_FIT_STREAMS=8 python3 tools/steady_state2.py 1 4000
} 2>&1 | grep -v "^$" | tee $OUT/steady.txt
FRANK_AMD_K2_RR=1 timeout 900 python3 bench.py --no-cpu-baseline --no-sharded > $OUT/bench_rr.json 2> $OUT/bench_rr.err

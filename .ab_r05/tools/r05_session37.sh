#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05s37; mkdir -p $OUT
{
echo "--- loaded, RR"; FRANK_AMD_K2_RR=1 python3 tools/k2_loaded.py 1 64 128 192 256 256
echo "--- loaded, default"; python3 tools/k2_loaded.py 1 64 128 192 256
echo "--- steady state, the forms in memory"; python3 tools/steady_state2.py 1 4000
echo "--- steady state, the matrix in registers"; FRANK_AMD_K2_RR=1 python3 tools/steady_state2.py 1 4000
FRANK_AMD_K2_RR=1 FRANK_AMD_FIT_SLOTS=300 python3 tools/steady_state2.py 1 4000
} 2>&1 | grep -v "^$" | tee $OUT/steady.txt

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05s34; mkdir -p $OUT
FRANK_AMD_K2_RR=1 timeout 600 python3 tools/rr_sweep.py 47 303 16 > $OUT/rr_sweep.txt 2>&1
echo "--- RR" > $OUT/loaded.txt
FRANK_AMD_K2_RR=1 timeout 600 python3 tools/k2_loaded.py 1 256 >> $OUT/loaded.txt 2>&1
FRANK_AMD_LIB=$PWD/frank_amd/libfrank_hip_timing.so FRANK_AMD_K2_RR=1 timeout 300 python3 tools/rr_trace.py > $OUT/rr_trace.txt 2>&1

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05s33; mkdir -p $OUT
FRANK_AMD_LIB=$PWD/frank_amd/libfrank_hip_timing.so FRANK_AMD_K2_RR=1 timeout 300 python3 tools/rr_trace.py > $OUT/rr_trace.txt 2>&1

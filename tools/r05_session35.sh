#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05s35; mkdir -p $OUT
FRANK_AMD_LIB=$PWD/frank_amd/libfrank_hip_rrsafe.so timeout 600 python3 tools/rr_one.py 303 > $OUT/rr_one.txt 2>&1

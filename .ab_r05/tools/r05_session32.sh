#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05s32; mkdir -p $OUT
timeout 600 python3 tools/rr_sweep.py 47 303 16 > $OUT/rr_sweep.txt 2>&1
echo "--- RR" > $OUT/loaded.txt
timeout 600 python3 tools/k2_loaded.py 1 64 256 >> $OUT/loaded.txt 2>&1
echo "--- DF" >> $OUT/loaded.txt
FRANK_AMD_K2_RR=0 timeout 600 python3 tools/k2_loaded.py 1 64 256 >> $OUT/loaded.txt 2>&1

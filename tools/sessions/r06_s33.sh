#!/bin/bash
# round 6, session 33: phases of the WIDE form (timing build), N = 640: distributed and local
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s33; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
{ FRANK_AMD_LIB=$ROOT/frank_amd/libfrank_hip_timing.so timeout 300 python3 tools/ln_n640.py 640 2>&1 | grep -v "$F" | head -3
  echo "--- FRANK_AMD_LN_CLUSTER_CHOL=0"
  FRANK_AMD_LN_CLUSTER_CHOL=0 FRANK_AMD_LIB=$ROOT/frank_amd/libfrank_hip_timing.so timeout 300 python3 tools/ln_n640.py 640 2>&1 | grep -v "$F" | head -3
} > $OUT/phases_wide2.txt
cat $OUT/phases_wide2.txt

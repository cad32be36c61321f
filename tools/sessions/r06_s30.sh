#!/bin/bash
# round 6, session 30: distributed Cholesky, the diagonal tile through LDS: shipped build (time, bits) and timing build (phases)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s30; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
{ echo "--- distributed"; timeout 120 python3 tools/ln_fullsize.py 1e7 linear 2>&1 | grep -v "$F"
  echo "--- FRANK_AMD_LN_CLUSTER_CHOL=0"; FRANK_AMD_LN_CLUSTER_CHOL=0 timeout 120 python3 tools/ln_fullsize.py 1e7 linear 2>&1 | grep -v "$F"
  echo "--- timing build"; FRANK_AMD_LIB=$ROOT/frank_amd/libfrank_hip_timing.so timeout 120 python3 tools/ln_fullsize.py 1e7 linear 2>&1 | grep -v "$F"
} > $OUT/ln.txt 2>&1
cat $OUT/ln.txt

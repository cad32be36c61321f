#!/bin/bash
# round 6, session 32: distributed Cholesky with paired device-scope loads: N = 300 (bits, time), N = 400 / 640, the batched sweep
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s32; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
{ echo "--- distributed"; timeout 120 python3 tools/ln_fullsize.py 1e7 linear 2>&1 | grep -v "$F"
  echo "--- FRANK_AMD_LN_CLUSTER_CHOL=0"; FRANK_AMD_LN_CLUSTER_CHOL=0 timeout 120 python3 tools/ln_fullsize.py 1e7 linear 2>&1 | grep -v "$F"
  for n in 400 640; do timeout 300 python3 tools/ln_n640.py $n 2>&1 | grep -v "$F"; FRANK_AMD_LN_CLUSTER_CHOL=0 timeout 300 python3 tools/ln_n640.py $n 2>&1 | grep -v "$F" | sed 's/^/   (CHOL=0) /'; done
  timeout 300 python3 tools/ln_batched64.py 2>&1 | grep -v "$F" | tail -3
} > $OUT/times.txt 2>&1
cat $OUT/times.txt

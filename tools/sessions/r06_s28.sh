#!/bin/bash
# round 6, session 28: the trailing tiles of the LogNormal Cholesky on the cluster's helpers (registers), same bits?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s28; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
{ for rep in 1 2 3; do
    echo "--- distributed Cholesky, process $rep"; timeout 120 python3 tools/ln_fullsize.py 1e7 linear 2>&1 | grep -v "$F"
  done
  echo "--- FRANK_AMD_LN_CLUSTER_CHOL=0"; FRANK_AMD_LN_CLUSTER_CHOL=0 timeout 120 python3 tools/ln_fullsize.py 1e7 linear 2>&1 | grep -v "$F"
} > $OUT/ln_fullsize.txt 2>&1
cat $OUT/ln_fullsize.txt
timeout 900 python3 -m pytest tests -m gpu -x -q -k "lognormal or LogNormal" 2>&1 | grep -v "$F" | tail -5

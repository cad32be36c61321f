#!/bin/bash
# round 6, session 36: the Newton directions of the WIDE form with their far products on the helpers: bits (helpers == local == one
# workgroup), the oracle, times at N = 400 / 640 and on the many-steps table
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s36; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python3 -m pytest tests -m gpu -x -q -k "cholesky_on_the_helpers or beyond_the_persistent" 2>&1 | grep -v "$F" | tail -30
{ for n in 400 640; do timeout 300 python3 tools/ln_n640.py $n 2>&1 | grep -v "$F"; FRANK_AMD_LN_CLUSTER_CHOL=0 timeout 300 python3 tools/ln_n640.py $n 2>&1 | grep -v "$F" | sed 's/^/   (CHOL=0) /'; done
  timeout 600 python3 tools/ln_wide_time.py 330 639 2>&1 | grep -v "$F"
} > $OUT/times.txt 2>&1
cat $OUT/times.txt

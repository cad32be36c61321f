#!/bin/bash
# round 6, session 31: the distributed Cholesky over the basis sizes (cluster of 8 / 3 == one workgroup, bit for bit; N <= 320 and the
# WIDE form), the WIDE form's times, the batched sweep, all LogNormal tests
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s31; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python3 tools/size_sweep_lognormal.py 2>&1 | grep -v "$F" | tail -32 > $OUT/size_sweep_lognormal.txt
tail -3 $OUT/size_sweep_lognormal.txt
timeout 900 python3 tools/size_sweep_ln_wide.py 2>&1 | grep -v "$F" | tail -42 > $OUT/size_sweep_ln_wide.txt
tail -2 $OUT/size_sweep_ln_wide.txt
{ for n in 400 640; do timeout 300 python3 tools/ln_n640.py $n 2>&1 | grep -v "$F"; FRANK_AMD_LN_CLUSTER_CHOL=0 timeout 300 python3 tools/ln_n640.py $n 2>&1 | grep -v "$F" | sed 's/^/   (CHOL=0) /'; done
  timeout 300 python3 tools/ln_batched64.py 2>&1 | grep -v "$F" | tail -4
  FRANK_AMD_LN_CLUSTER_CHOL=0 timeout 300 python3 tools/ln_batched64.py 2>&1 | grep -v "$F" | tail -4 | sed 's/^/   (CHOL=0) /'
} > $OUT/times.txt 2>&1
cat $OUT/times.txt
timeout 1200 python3 -m pytest tests -m gpu -x -q -k "lognormal or LogNormal or sweep" 2>&1 | grep -v "$F" | tail -4

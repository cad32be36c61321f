#!/bin/bash
# round 6, session 17: where the WIDE LogNormal kernel spends a pass (timing build), N = 640 on the bench's workload and N = 639 on the
# many-steps workload
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s17; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
export FRANK_AMD_LIB=$ROOT/frank_amd/libfrank_hip_timing.so
{ timeout 300 python3 tools/ln_n640.py 640 2>&1 | grep -v "$F"
  timeout 300 python3 tools/ln_n640.py 400 2>&1 | grep -v "$F"
  timeout 300 python3 tools/ln_wide_time.py 639 2>&1 | grep -v "$F"
} > $OUT/phases_wide.txt 2>&1
cat $OUT/phases_wide.txt

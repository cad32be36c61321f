#!/bin/bash
# round 6, session 29: the distributed Cholesky under the phase timers (timing build): where the first fit of a process loses 2.7 s
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s29; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
FRANK_AMD_LIB=$ROOT/frank_amd/libfrank_hip_timing.so timeout 120 python3 tools/ln_fullsize.py 1e7 linear > $OUT/ln_timing.txt 2>&1
grep -v "$F" $OUT/ln_timing.txt

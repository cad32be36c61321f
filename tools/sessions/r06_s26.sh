#!/bin/bash
# round 6, session 26: the in-kernel timeline of a cluster pass (timing build) once more
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s26; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
FRANK_AMD_LIB=$ROOT/frank_amd/libfrank_hip_timing.so timeout 300 python3 tools/k2_quick_h.py 300 2>&1 | grep -v "$F" > $OUT/k2_timeline.txt
cat $OUT/k2_timeline.txt

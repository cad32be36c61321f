#!/bin/bash
# round 6, session 5: phases of the LogNormal fit with the new Cholesky (timing build), same-bits check of the shipped build
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s05; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
timeout 300 python3 tools/ln_fullsize.py 1e7 linear 2>&1 | grep -v "$F" > $OUT/ln_fullsize.txt
FRANK_AMD_LIB=$ROOT/frank_amd/libfrank_hip_timing.so timeout 600 python3 tools/ln_phases.py > $OUT/ln_phases.out 2> $OUT/ln_phases.txt
cat $OUT/ln_fullsize.txt; grep -v "$F" $OUT/ln_phases.txt

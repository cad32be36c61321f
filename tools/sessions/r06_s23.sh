#!/bin/bash
# round 6, session 23: one pipeline or two contexts taking turns (20-step region, steady state, the distinct-datasets sweep)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s23; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python3 tools/two_contexts.py 1500 2>&1 | grep -v "$F" > $OUT/two_contexts.txt
cat $OUT/two_contexts.txt

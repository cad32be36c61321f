#!/bin/bash
# round 6, session 34: the whole GPU suite and the smoke on the library with the distributed Cholesky
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s34; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -8 > $OUT/pytest_gpu.txt
tail -3 $OUT/pytest_gpu.txt
timeout 300 python3 __graft_entry__.py smoke 2>&1 | grep -v "$F" | tail -4

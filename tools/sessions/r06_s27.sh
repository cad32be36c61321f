#!/bin/bash
# round 6, session 27: the oracle as the referee of the WIDE LogNormal kernel at N = 639 too
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s27; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
timeout 1500 python3 -m pytest tests -m gpu -x -q --durations=3 -k "beyond_the_persistent" 2>&1 | grep -v "$F" | tail -25 > $OUT/pytest_ln.txt
cat $OUT/pytest_ln.txt

#!/bin/bash
# round 6, session 9: the staged LogNormal sweep (pause / resume, several clusters in one launch): its test, the LogNormal tests, the
# bench's 64-point leg before / after
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s09; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python3 -m pytest tests -m gpu -x -q -k "lognormal or LogNormal" 2>&1 | grep -v "$F" | tail -15 > $OUT/pytest_lognormal.txt
timeout 600 python3 tools/ln_batched64.py > $OUT/ln_batched64.txt 2>&1
tail -15 $OUT/pytest_lognormal.txt; grep -v "$F" $OUT/ln_batched64.txt | tail

#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s11
mkdir -p $OUT
cd $ROOT
timeout 1500 python3 -m pytest tests -m gpu -x -q --timeout 900 -k "staged or sweep or batched or deferred" 2>&1 | tail -8 | tee $OUT/tests.txt
{
FRANK_AMD_SWEEP_CAP=0 python3 tools/sweep512_tune.py
for cap in 400 500 640 700 800; do FRANK_AMD_SWEEP_CAP=$cap python3 tools/sweep512_tune.py; done
for k2 in 16 24 32; do FRANK_AMD_SWEEP_CAP=640 FRANK_AMD_SWEEP_STAGE2_CLUSTERS=$k2 python3 tools/sweep512_tune.py; done
FRANK_AMD_SWEEP_CAP=640 FRANK_AMD_K2_PAIR=0 python3 tools/sweep512_tune.py
} 2>&1 | grep -v "^$" | tee $OUT/sweep512.txt

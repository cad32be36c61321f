#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s9
mkdir -p $OUT
cd $ROOT
{
python3 tools/sweep512_tune.py
FRANK_AMD_K2_DEFER=0 python3 tools/sweep512_tune.py
for k in 0 7 8 12 16 20; do
  for pr in 0 1; do FRANK_AMD_SWEEP_CLUSTERS=$k FRANK_AMD_K2_PAIR=$pr python3 tools/sweep512_tune.py; done
done
FRANK_AMD_SWEEP_CLUSTERS=8 FRANK_AMD_K2_CLUSTER=8 FRANK_AMD_K2_PAIR=1 python3 tools/sweep512_tune.py
FRANK_AMD_SWEEP_CLUSTERS=10 FRANK_AMD_K2_CLUSTER=8 FRANK_AMD_K2_PAIR=1 python3 tools/sweep512_tune.py
} 2>&1 | grep -v "^$" | tee $OUT/sweep512.txt

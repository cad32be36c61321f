#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s16
mkdir -p $OUT
cd $ROOT
timeout 900 python3 -m pytest tests -m gpu -x -q --timeout 900 -k "bucket_tables or staged or sweep or batched" 2>&1 | tail -8 | tee $OUT/tests.txt
{
for cap in -1 0 1000; do BENCH_TABLE=1 PRINT_ITS=${PI:-} FRANK_AMD_SWEEP_TRACE=1 FRANK_AMD_SWEEP_CAP=$cap python3 tools/sweep512_tune.py; done
for l in 24 32 40; do FRANK_AMD_SWEEP_LEFT=$l BENCH_TABLE=1 FRANK_AMD_SWEEP_TRACE=1 python3 tools/sweep512_tune.py; done
for cap in -1 0; do FRANK_AMD_SWEEP_TRACE=1 FRANK_AMD_SWEEP_CAP=$cap python3 tools/sweep512_tune.py; done
} 2>&1 | grep -v "^$" | tee $OUT/sweep512.txt

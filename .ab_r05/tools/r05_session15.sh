#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s15
mkdir -p $OUT
cd $ROOT
timeout 900 python3 -m pytest tests -m gpu -x -q --timeout 900 -k "bucket_tables or moment_path_equals or test_fit_N300_1e7" 2>&1 | tail -8 | tee $OUT/tests.txt
{
for cap in 0 700 800 900 1000 1100; do BENCH_TABLE=1 PRINT_ITS=${PI:-} FRANK_AMD_SWEEP_TRACE=1 FRANK_AMD_SWEEP_CAP=$cap python3 tools/sweep512_tune.py; done
BENCH_TABLE=1 PRINT_ITS=1 FRANK_AMD_SWEEP_CAP=800 python3 tools/sweep512_tune.py | tail -1 | cut -c1-1500
} 2>&1 | grep -v "^$" | tee $OUT/sweep512.txt

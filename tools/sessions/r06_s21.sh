#!/bin/bash
# round 6, session 21: the stragglers of a pipeline's launch pause and join the next launch: the new test, the pipeline tests, the
# distinct-datasets sweep with and without it
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s21; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python3 -m pytest tests -m gpu -x -q -k "pipeline or pipelined or staged or sweep or resume or pause" 2>&1 | grep -v "$F" | tail -8 > $OUT/pytest_pipeline.txt
tail -8 $OUT/pytest_pipeline.txt
timeout 600 python3 tools/sweep512_distinct.py 2 2>&1 | grep -v "$F" > $OUT/sweep512_distinct.txt
cat $OUT/sweep512_distinct.txt

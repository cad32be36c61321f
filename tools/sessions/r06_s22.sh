#!/bin/bash
# round 6, session 22: the pipeline picks the stream that will be free first; the distinct-datasets sweep with its longest fits on clusters
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s22; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python3 -m pytest tests -m gpu -x -q -k "pipeline or pipelined or staged or sweep" 2>&1 | grep -v "$F" | tail -4 > $OUT/pytest_pipeline.txt
tail -3 $OUT/pytest_pipeline.txt
timeout 600 python3 tools/sweep512_distinct.py 2 2>&1 | grep -v "$F" > $OUT/sweep512_distinct.txt
cat $OUT/sweep512_distinct.txt
{ echo "--- steady state, ring of 4, range cache off"; timeout 300 python3 tools/steady_state.py 2000 --distinct 4 2>&1 | grep -v "$F" | tail -1 | cut -c1-160
  echo "--- steady state, one table"; timeout 300 python3 tools/steady_state.py 2000 2>&1 | grep -v "$F" | tail -1 | cut -c1-160; } > $OUT/steady.txt 2>&1
cat $OUT/steady.txt

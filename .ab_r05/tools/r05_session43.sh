#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05s43; mkdir -p $OUT
{
for spec in "240 64 6" "240 32 6" "240 16 8" "240 96 6" "240 64 4" "240 64 8" "200 64 6" "220 48 6" "256 64 6" "256 32 8" "192 32 6"; do
  set -- $spec
  echo -n "slots=$1 batch=$2 streams=$3: "
  FRANK_AMD_FIT_SLOTS=$1 FRANK_AMD_FIT_BATCH=$2 FRANK_AMD_FIT_STREAMS=$3 python3 tools/steady_state2.py 1 4000 2>&1 | grep contexts | sed 's/.*each: //'
done
} | tee $OUT/geometry_rr.txt

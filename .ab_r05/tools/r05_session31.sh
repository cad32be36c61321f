#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05s31; mkdir -p $OUT
{
python3 tools/steady_state2.py 1
for spec in "2 120 3" "2 120 2" "2 160 3" "2 96 2" "3 80 2" "3 80 1" "4 60 1" "2 240 3"; do
  set -- $spec
  FRANK_AMD_FIT_SLOTS=$2 FRANK_AMD_FIT_STREAMS=$3 python3 tools/steady_state2.py $1
done
FRANK_AMD_K2_PAIR=1 FRANK_AMD_FIT_SLOTS=120 FRANK_AMD_FIT_STREAMS=3 python3 tools/steady_state2.py 2
} 2>&1 | grep contexts | tee $OUT/two_contexts.txt
FRANK_AMD_LIB=$PWD/frank_amd/libfrank_hip_timing.so timeout 300 python3 tools/k2_quick_h.py 300 2>&1 | tail -75 > $OUT/cluster_timeline.txt

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05s48; mkdir -p $OUT
{
for b in 512 640 768 896 512 768; do
  echo -n "P2 workgroups=$b (8 waves): "
  FRANK_AMD_K1_WPB=8 FRANK_AMD_K1_BLOCKS=$b python3 tools/steady_state2.py 1 4000 2>&1 | grep contexts | sed 's/.*each: //'
  FRANK_AMD_K1_WPB=8 FRANK_AMD_K1_BLOCKS=$b python3 tools/k1_pass.py 1e7 300 20 2>&1 | grep "ms per pass"
done
} | tee $OUT/binning_workgroups.txt

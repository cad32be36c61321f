#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05s45; mkdir -p $OUT
{
for spec in "0 0" "8 512" "8 768" "8 1024" "4 1024" "4 2048" "16 256"; do
  set -- $spec
  echo -n "wpb=$1 blocks=$2: "
  FRANK_AMD_K1_WPB=$1 FRANK_AMD_K1_BLOCKS=$2 python3 tools/steady_state2.py 1 4000 2>&1 | grep contexts | sed 's/.*each: //'
  FRANK_AMD_K1_WPB=$1 FRANK_AMD_K1_BLOCKS=$2 python3 tools/k1_pass.py 1e7 300 5 2>&1 | tail -2
done
} | tee $OUT/binning_geometry.txt

#!/bin/bash
# Round 5, session 5: the pipeline's launch geometry again, now that a full device pays (deferred + paired kernel).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s5
mkdir -p $OUT
cd $ROOT
run () { python3 tools/steady_state.py 2>&1 | tail -1 | sed -e "s/, 'steps.*//" ; }
{
for pair in auto 1 0; do
for spec in "240 64 6" "255 64 6" "255 64 4" "255 85 4" "255 85 3" "255 128 3" "255 48 6" "255 32 8"; do
  set -- $spec
  echo -n "pair=$pair slots=$1 batch=$2 streams=$3: "
  if [ $pair = auto ]; then FRANK_AMD_FIT_SLOTS=$1 FRANK_AMD_FIT_BATCH=$2 FRANK_AMD_FIT_STREAMS=$3 run
  else FRANK_AMD_K2_PAIR=$pair FRANK_AMD_FIT_SLOTS=$1 FRANK_AMD_FIT_BATCH=$2 FRANK_AMD_FIT_STREAMS=$3 run; fi
done
done
echo -n "kernel of rounds 2-4, 240/64/6: "; FRANK_AMD_K2_DEFER=0 run
} 2>&1 | tee $OUT/geometry.txt

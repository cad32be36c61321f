#!/bin/bash
# Round 5, session 7: deep launch queues (more outstanding fits than compute units: the next launch of a stream is already
# waiting when the one before ends) + the new parity tests.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s7
mkdir -p $OUT
cd $ROOT
timeout 2400 python3 -m pytest tests -m gpu -x -q --timeout 1500 -k "xwide or two_processes or interpolation or deferred or prepass or hist" 2>&1 | tail -8 | tee $OUT/tests.txt
run () { python3 tools/steady_state.py 2>&1 | tail -1 | sed -e "s/.*'fits_per_s': //" -e "s/, 'steps.*//" ; }
{
for pair in 0 1; do
for spec in "240 64 6" "320 64 3" "384 64 3" "448 64 3" "320 64 4" "384 64 4" "448 64 4" "512 64 4" "384 48 4" "448 48 5" "512 96 3"; do
  set -- $spec
  echo -n "pair=$pair slots=$1 batch=$2 streams=$3: "
  FRANK_AMD_K2_PAIR=$pair FRANK_AMD_FIT_SLOTS=$1 FRANK_AMD_FIT_BATCH=$2 FRANK_AMD_FIT_STREAMS=$3 run
done
done
} 2>&1 | tee $OUT/geometry.txt

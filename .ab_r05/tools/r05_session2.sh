#!/bin/bash
# Round 5, second GPU session: the deferred trailing update.  bits first, then time alone and loaded, then the steady state.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05s2}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
echo "== bits" | tee $OUT/bits.txt
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "deferred or left_looking or cluster_mode_equals or diagonal_tile" 2>&1 | tail -5 | tee -a $OUT/bits.txt
timeout 600 python3 tools/defer_sweep.py 47 319 2 2>&1 | tail -8 | tee -a $OUT/bits.txt
export FRANK_AMD_SWEEP_NO_CLUSTERS=1
for d in 1 0; do
  echo "== clocks, FRANK_AMD_K2_DEFER=$d" | tee -a $OUT/clocks.txt
  FRANK_AMD_K2_DEFER=$d timeout 600 python3 tools/k2_loaded.py --steady 1 64 128 192 256 2>&1 | tee -a $OUT/clocks.txt
done
echo "== counters, 256 loops resident, deferred"
bash tools/k2_batch_pmc.sh $TAG 256 2>&1 | tail -3
cp $OUT/prof_k2_batch256/pmc.json $OUT/pmc_fit_loop_256_deferred.json 2>/dev/null
ls $OUT

#!/bin/bash
# round 6, session 11: A/B on ONE box -- the tree of the end of round 5 (.ab_r05/, built here from commit 30dd7cb) against this tree:
# the pipeline at steady state (caches on: the figure of round 5) and the driver's 20-step region, twice each, alternating
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s11; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
{ for rep in 1 2; do
    echo "--- round-5 tree, steady state (one table, caches on)"; ( cd .ab_r05 && timeout 300 python3 tools/steady_state.py 2000 2>&1 | grep -v "$F" | tail -1 | cut -c1-200 )
    echo "--- this tree, steady state (one table, caches on)"; timeout 300 python3 tools/steady_state.py 2000 2>&1 | grep -v "$F" | tail -1 | cut -c1-200
    echo "--- this tree, steady state (ring of 4 table objects, range cache off, look-ahead)"; timeout 300 python3 tools/steady_state.py 2000 --distinct 4 2>&1 | grep -v "$F" | tail -1 | cut -c1-200
  done
  echo "--- round-5 tree, bench --steps 20 --no-extras"; ( cd .ab_r05 && timeout 300 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | cut -c1-330 )
  echo "--- this tree, bench --steps 20 --no-extras"; timeout 300 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | cut -c1-330
} > $OUT/ab_round5_against_round6.txt 2>&1
cat $OUT/ab_round5_against_round6.txt

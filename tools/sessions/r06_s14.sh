#!/bin/bash
# round 6, session 14: A/B on one box -- the library of the last commit (range look-ahead, two looks at (u, v)) against the one-look build
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s14; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
{ for rep in 1 2; do
    for lib in libfrank_hip_a.so libfrank_hip.so; do
      echo "--- $lib: steady state, ring of 4, range cache off, look-ahead"; FRANK_AMD_LIB=$ROOT/frank_amd/$lib timeout 300 python3 tools/steady_state.py 2000 --distinct 4 2>&1 | grep -v "$F" | tail -1 | cut -c1-120
      echo "--- $lib: 20-step region"; FRANK_AMD_LIB=$ROOT/frank_amd/$lib timeout 300 python3 tools/submit_phase_ring.py 20 2>&1 | grep "look-ahead\|caches on" | tail -3 | cut -c1-200
    done
  done; } > $OUT/ab_one_look.txt 2>&1
cat $OUT/ab_one_look.txt

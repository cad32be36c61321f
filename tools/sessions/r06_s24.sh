#!/bin/bash
# round 6, session 24: the final library -- the whole GPU suite, the smoke, then the profile set of the round (tools/profile_r06.sh)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -12 > $OUT/pytest_gpu.txt
tail -3 $OUT/pytest_gpu.txt
timeout 300 python3 __graft_entry__.py smoke 2>&1 | grep -v "$F" | tail -8 > $OUT/smoke.txt
tail -3 $OUT/smoke.txt
timeout 2400 bash tools/profile_r06.sh r06 > $OUT/profile_r06.log 2>&1
tail -12 $OUT/profile_r06.log | cut -c1-250; cat $OUT/steady_state_under_rocprof.txt | tail -1 | cut -c1-200; cat $OUT/library.txt

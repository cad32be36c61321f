#!/bin/bash
# round 6, session 37: the WIDE form's chains on a diagonal block staged in LDS: tests, times, phases
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s37; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python3 -m pytest tests -m gpu -x -q -k "cholesky_on_the_helpers or beyond_the_persistent" 2>&1 | grep -v "$F" | tail -4
{ for n in 400 640; do timeout 300 python3 tools/ln_n640.py $n 2>&1 | grep -v "$F"; done
  timeout 600 python3 tools/ln_wide_time.py 330 639 2>&1 | grep -v "$F"
  FRANK_AMD_LIB=$ROOT/frank_amd/libfrank_hip_timing.so timeout 300 python3 tools/ln_n640.py 640 2>&1 | grep -v "$F" | head -2
} > $OUT/times.txt 2>&1
cat $OUT/times.txt

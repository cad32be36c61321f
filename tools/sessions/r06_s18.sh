#!/bin/bash
# round 6, session 18: the WIDE form's Newton directions with the chain of a block under the products of the next one
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s18; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python3 -m pytest tests -m gpu -x -q -k "beyond_the_persistent" 2>&1 | grep -v "$F" | tail -5 > $OUT/pytest_ln.txt
tail -3 $OUT/pytest_ln.txt
{ for n in 400 640; do timeout 300 python3 tools/ln_n640.py $n 2>&1 | grep -v "$F"; done
  timeout 600 python3 tools/ln_wide_time.py 330 639 2>&1 | grep -v "$F"
} > $OUT/ln_wide_time.txt 2>&1
cat $OUT/ln_wide_time.txt

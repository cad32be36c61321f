#!/bin/bash
# round 6, session 42: soak of the distributed Cholesky: the same fit 150-300 times at three sizes, alone and with a second process
# doing the same on the device
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s42; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
{ timeout 600 python3 tools/soak_ln_chol.py 300 300 2>&1 | grep -v "$F"
  timeout 600 python3 tools/soak_ln_chol.py 208 300 2>&1 | grep -v "$F"
  timeout 600 python3 tools/soak_ln_chol.py 400 150 2>&1 | grep -v "$F"
  timeout 600 python3 tools/soak_ln_chol.py 640 60 2>&1 | grep -v "$F"
  echo "--- two processes at once"
  timeout 600 python3 tools/soak_ln_chol.py 300 150 > $OUT/p1.txt 2>&1 &
  timeout 600 python3 tools/soak_ln_chol.py 304 150 > $OUT/p2.txt 2>&1 &
  wait
  grep -v "$F" $OUT/p1.txt; grep -v "$F" $OUT/p2.txt
} > $OUT/soak.txt 2>&1
cat $OUT/soak.txt

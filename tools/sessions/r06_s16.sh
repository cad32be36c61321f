#!/bin/bash
# round 6, session 16: the WIDE LogNormal kernel with its vectors back in LDS (the Cholesky's panel over them): the LogNormal tests,
# whole fits at N = 330..640 against the host-driven route, then the whole GPU suite and the smoke
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s16; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python3 -m pytest tests -m gpu -x -q -k "lognormal or LogNormal" 2>&1 | grep -v "$F" | tail -25 > $OUT/pytest_ln.txt
tail -8 $OUT/pytest_ln.txt
{ echo "--- persistent kernel, WIDE form"; timeout 600 python3 tools/ln_wide_time.py 330 400 512 639 2>&1 | grep -v "$F"
  for n in 400 640; do timeout 300 python3 tools/ln_n640.py $n 2>&1 | grep -v "$F"; done
  echo "--- host-driven route"; FRANK_AMD_LN_WIDE=host timeout 300 python3 tools/ln_n640.py 640 2>&1 | grep -v "$F"
} > $OUT/ln_wide_time.txt 2>&1
cat $OUT/ln_wide_time.txt
timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -15 > $OUT/pytest_gpu.txt
tail -6 $OUT/pytest_gpu.txt
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v "$F" | tail -3

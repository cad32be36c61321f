#!/bin/bash
# round 6, session 15: the persistent LogNormal kernel in its WIDE form (320 < N <= 639): the oracle's test at N = 330 / 400, whole fits
# timed against the host-driven route of round 4 (FRANK_AMD_LN_WIDE=host), the N <= 320 tests untouched
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s15; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python3 -m pytest tests -m gpu -x -q -k "beyond_the_persistent or lognormal_wide or test_lognormal_odd or test_lognormal_map_model" 2>&1 | grep -v "$F" | tail -25 > $OUT/pytest_wide.txt
{ echo "--- persistent kernel, WIDE form"; timeout 600 python3 tools/ln_wide_time.py 330 400 512 639 2>&1 | grep -v "$F"
  echo "--- host-driven route (FRANK_AMD_LN_WIDE=host)"; FRANK_AMD_LN_WIDE=host timeout 600 python3 tools/ln_wide_time.py 330 400 512 639 2>&1 | grep -v "$F"
  echo "--- one workgroup (FRANK_AMD_LN_CLUSTER=1), WIDE form"; FRANK_AMD_LN_CLUSTER=1 timeout 600 python3 tools/ln_wide_time.py 400 2>&1 | grep -v "$F"
} > $OUT/ln_wide_time.txt 2>&1
tail -25 $OUT/pytest_wide.txt; cat $OUT/ln_wide_time.txt

#!/bin/bash
# round 6, session 8: Newton directions through the inverses of the 64 x 64 diagonal blocks (block_solve_li): time, counters, phases,
# the LogNormal tests (the arithmetic of a direction changes: not the bits of rounds 3-5 any more)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s08; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
python3 -c "import sys; sys.path.insert(0,'.'); from frank_amd import _lib as L; print(L.lib.fh_version().decode())" > $OUT/library.txt
{ for mode in linear reference; do
    echo "--- cluster, $mode"; timeout 300 python3 tools/ln_fullsize.py 1e7 $mode 2>&1 | grep -v "$F"
    echo "--- one workgroup, $mode"; FRANK_AMD_LN_CLUSTER=1 timeout 300 python3 tools/ln_fullsize.py 1e7 $mode 2>&1 | grep -v "$F"
  done; } > $OUT/ln_fullsize.txt 2>&1
FRANK_AMD_LIB=$ROOT/frank_amd/libfrank_hip_timing.so timeout 600 python3 tools/ln_phases.py > $OUT/ln_phases.out 2> $OUT/ln_phases.txt
timeout 1200 python3 -m pytest tests -m gpu -q -k "lognormal or LogNormal" 2>&1 | grep -v "$F" | tail -25 > $OUT/pytest_lognormal.txt
cat $OUT/ln_fullsize.txt; grep "ln timing\|    [0-9]" $OUT/ln_phases.txt | head -8; grep "Cholesky" $OUT/ln_phases.out | head -2; tail -25 $OUT/pytest_lognormal.txt

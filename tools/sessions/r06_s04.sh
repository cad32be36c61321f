#!/bin/bash
# round 6, session 4: the LogNormal Cholesky on packed tiles with one barrier per step -- same bits as the routine of rounds 3-5?
# (libfrank_hip_old.so = this tree with lognormal.hip of HEAD), time, phases, the LogNormal tests
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s04; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
python3 -c "import sys; sys.path.insert(0,'.'); from frank_amd import _lib as L; print(L.lib.fh_version().decode())" > $OUT/library.txt
{ for mode in linear reference; do
    echo "--- OLD (rounds 3-5 Cholesky), $mode"; FRANK_AMD_LIB=$ROOT/frank_amd/libfrank_hip_old.so timeout 300 python3 tools/ln_fullsize.py 1e7 $mode 2>&1 | grep -v "$F"
    echo "--- NEW, $mode"; timeout 300 python3 tools/ln_fullsize.py 1e7 $mode 2>&1 | grep -v "$F"
    echo "--- NEW, one workgroup, $mode"; FRANK_AMD_LN_CLUSTER=1 timeout 300 python3 tools/ln_fullsize.py 1e7 $mode 2>&1 | grep -v "$F"
  done; } > $OUT/ln_fullsize_old_new.txt 2>&1
FRANK_AMD_LIB=$ROOT/frank_amd/libfrank_hip_timing.so timeout 600 python3 tools/ln_phases.py > $OUT/ln_phases.out 2> $OUT/ln_phases.txt
timeout 1200 python3 -m pytest tests -m gpu -x -q -k "lognormal or LogNormal or fused" 2>&1 | grep -v "$F" | tail -8 > $OUT/pytest_lognormal.txt
cat $OUT/ln_fullsize_old_new.txt; grep -v "$F" $OUT/ln_phases.txt; cat $OUT/pytest_lognormal.txt

#!/bin/bash
# the regression sweeps of the round once more, on the final library
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05s56; mkdir -p $OUT
python3 -c "import sys; sys.path.insert(0,'.'); from frank_amd import _lib as L; print(L.lib.fh_version().decode())" > $OUT/library.txt
timeout 1500 python3 tools/size_sweep_cluster.py 2>&1 | tail -6 > $OUT/size_sweep_cluster.txt
{ echo "--- tools/defer_sweep.py 47 319 1 (FRANK_AMD_K2_RR=0: the deferred form against the kernel of rounds 2-4)"; FRANK_AMD_K2_RR=0 timeout 900 python3 tools/defer_sweep.py 47 319 1 2>&1 | tail -3
  echo "--- tools/hyper_sweep_cluster.py (clusters against one workgroup, the form by load)"; timeout 900 python3 tools/hyper_sweep_cluster.py 2>&1 | tail -4; } > $OUT/deferred_and_hyper_sweeps.txt

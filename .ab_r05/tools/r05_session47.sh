#!/bin/bash
# regression sweeps of the register-resident fit loop on the final binary
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05s47; mkdir -p $OUT
{
echo "--- every basis size 47 ... 319, the register-resident form against the forms in memory (tools/rr_sweep.py 47 319 1)"
timeout 1500 python3 tools/rr_sweep.py 47 319 1 2>&1 | tail -3
echo "--- hyper-parameter extremes at N = 130, 300 (400: outside the form), a cluster of workgroups against ONE workgroup in the register-resident form (FRANK_AMD_K2_RR=1 tools/hyper_sweep_cluster.py)"
FRANK_AMD_K2_RR=1 timeout 1500 python3 tools/hyper_sweep_cluster.py 2>&1 | tail -4
} > $OUT/rr_sweeps.txt 2>&1

#!/bin/bash
# round 6, session 19: the WIDE LogNormal kernel over 40 basis sizes (cluster == one workgroup, both against the host-driven route)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s19; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
timeout 1500 python3 tools/size_sweep_ln_wide.py 2>&1 | grep -v "$F" > $OUT/size_sweep_ln_wide.txt
tail -45 $OUT/size_sweep_ln_wide.txt

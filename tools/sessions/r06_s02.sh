#!/bin/bash
# round 6, session 2: the fused form of the moments pre-pass against the sorted one (tools/k1_fused.py)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r06s02; mkdir -p $OUT
python3 -c "import sys; sys.path.insert(0,'.'); from frank_amd import _lib as L; print(L.lib.fh_version().decode())" > $OUT/library.txt
timeout 600 python3 tools/k1_fused.py 1e7 300 50 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > $OUT/k1_fused.txt
timeout 600 python3 tools/k1_fused.py 1e7 300 50 2.0 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" > $OUT/k1_fused_stretch2.txt
cat $OUT/k1_fused.txt $OUT/k1_fused_stretch2.txt

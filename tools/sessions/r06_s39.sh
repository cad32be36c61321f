#!/bin/bash
# round 6, session 39: the LogNormal fit on a context that has run the Normal pipeline (bench.py's): why 0.57 s there and 0.35 s alone
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
echo "--- after the pipeline and the from-host leg"; timeout 300 python3 tools/ln_after_pipeline.py --from-host 2>&1 | grep -v "$F" | tail -3
echo "--- the same, CHOL=0"; FRANK_AMD_LN_CLUSTER_CHOL=0 timeout 300 python3 tools/ln_after_pipeline.py --from-host 2>&1 | grep -v "$F" | tail -3

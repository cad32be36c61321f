#!/bin/bash
# round 6, session 35: the new equality test of the distributed Cholesky
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python3 -m pytest tests -m gpu -x -q -k "cholesky_on_the_helpers" 2>&1 | grep -v "$F" | tail -6

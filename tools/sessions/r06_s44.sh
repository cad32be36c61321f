#!/bin/bash
# round 6, session 44: the GPU suite with its files in the other order (an order-dependent fault was found this way in round 5)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
timeout 2400 python3 -m pytest $(ls tests/test_*.py | sort -r) -m gpu -x -q 2>&1 | grep -v "$F" | tail -4

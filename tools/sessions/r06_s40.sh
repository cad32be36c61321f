#!/bin/bash
# round 6, session 40: the substitution chains of N <= 320 with two groups of columns in flight
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
timeout 120 python3 tools/ln_fullsize.py 1e7 linear 2>&1 | grep -v "$F" | tail -2
timeout 120 python3 tools/ln_fullsize.py 1e7 reference 2>&1 | grep -v "$F" | tail -1
timeout 900 python3 -m pytest tests -m gpu -x -q -k "lognormal or LogNormal" 2>&1 | grep -v "$F" | tail -3

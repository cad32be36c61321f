#!/bin/bash
# round 6, session 43: fits from host arrays back to back, every result checked (tables are freed behind their fits while fit loops run)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
timeout 600 python3 tools/from_host_phases.py 192 24 24 8 2>&1 | grep -v "$F" | tail -4
timeout 600 python3 tools/from_host_phases.py 96 8 8 2 2>&1 | grep -v "$F" | tail -4

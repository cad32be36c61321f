#!/bin/bash
# round 6, session 20: where the host's time goes in the fits from host arrays back to back
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s20; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
{ for args in "96 12 12 0" "96 16 16 4" "96 16 16 2" "96 16 16 1" "96 24 24 8"; do timeout 300 python3 tools/from_host_phases.py $args 2>&1 | grep -v "$F"; done
} > $OUT/from_host_phases.txt 2>&1
cat $OUT/from_host_phases.txt
timeout 900 python3 -m pytest tests -m gpu -x -q -k "upload or vis or map or pipeline or cache or bootstrap or mapping" 2>&1 | grep -v "$F" | tail -4

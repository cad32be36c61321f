#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s12
mkdir -p $OUT
cd $ROOT
timeout 1500 python3 -m pytest tests -m gpu -x -q --timeout 900 -k "staged or sweep or batched or cluster_mode_equals" 2>&1 | tail -6 | tee $OUT/tests.txt
{
for cap in 640 800 900 1000 1200; do FRANK_AMD_SWEEP_TRACE=1 FRANK_AMD_SWEEP_CAP=$cap python3 tools/sweep512_tune.py; done
} 2>&1 | grep -v "^$" | tee $OUT/sweep512.txt
echo "== cluster pass" | tee $OUT/cluster.txt
timeout 600 python3 tools/k2_cluster.py 2>&1 | tail -12 | tee -a $OUT/cluster.txt
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench20.json 2> $OUT/bench20.err
python3 -c "
import json; d=json.loads(open('$OUT/bench20.json').read().strip().split('\n')[-1]); print('value', d['value'], {k: v for k, v in d['breakdown_ms'].items() if k != 'note'})" | tee -a $OUT/cluster.txt

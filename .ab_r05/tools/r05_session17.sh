#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s17
mkdir -p $OUT
cd $ROOT
timeout 2400 python3 -m pytest tests -m gpu -x -q --timeout 1500 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 - <<'PY' | tee $OUT/summary.txt
import json
d = json.loads(open("gpurun_out/r05s17/bench.json").read().strip().split("\n")[-1])
ex = d["extra"]
print("value", d["value"], d["config"]["iterations_match_the_reference"])
for k in ("steady_state", "distinct_tables", "device_full", "from_host_arrays", "sweep512", "wide_uv", "lognormal_batched64"):
    e = ex.get(k, {})
    print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in e.items() if isinstance(vv, (int, float, list))}, e.get("error"))
PY
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-extras > $OUT/bench20.json 2> $OUT/bench20.err
python3 -c "
import json; d=json.loads(open('$OUT/bench20.json').read().strip().split('\n')[-1]); print('steps 20: value', d['value'], 'cpu', d.get('cpu_baseline',{}).get('value'))" | tee -a $OUT/summary.txt

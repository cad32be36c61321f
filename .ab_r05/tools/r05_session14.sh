#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s14
mkdir -p $OUT
cd $ROOT
timeout 1500 python3 -m pytest tests -m gpu -x -q --timeout 900 -k "bucket_tables or moment_path or wide or prepass or predict or fit_N300_1e7" 2>&1 | tail -8 | tee $OUT/tests.txt
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 - <<'PY' | tee $OUT/summary.txt
import json
d = json.loads(open("gpurun_out/r05s14/bench.json").read().strip().split("\n")[-1])
ex = d["extra"]
print("value", d["value"])
for k in ("steady_state", "device_full", "from_host_arrays", "sweep512", "wide_uv"):
    e = ex.get(k, {})
    print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in e.items() if isinstance(vv, (int, float, list))}, e.get("error"))
PY

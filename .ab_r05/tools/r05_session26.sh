#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05s26; mkdir -p $OUT
timeout 600 python3 -m pytest tests -m gpu -x -q --timeout 500 -k "(staged_sweep_equals and 130-200) or lognormal_fit_N40" 2>&1 | tail -2
timeout 2400 python3 -m pytest tests -m gpu -x -q --timeout 1500 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt
timeout 1500 python3 -m pytest tests -m gpu -x -q --timeout 900 -k "lognormal or staged or sweep" 2>&1 | tail -2 | tee -a $OUT/pytest_gpu.txt
timeout 900 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 - <<'PY' | tee $OUT/summary.txt
import json
d = json.loads(open("gpurun_out/r05s26/bench.json").read().strip().split("\n")[-1])
ex = d["extra"]
print("value", d["value"], d["config"]["iterations_match_the_reference"])
for k in ("steady_state", "distinct_tables", "device_full", "from_host_arrays", "sweep512", "wide_uv", "lognormal_batched64", "lognormal_fullsize"):
    e = ex.get(k, {})
    print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in e.items() if isinstance(vv, (int, float, list))}, e.get("error"))
PY
FRANK_AMD_FIT_EARLY=1 timeout 600 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('FIT_EARLY=1 steps 20: value', d['value'])" | tee -a $OUT/summary.txt
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('default steps 20: value', d['value'])" | tee -a $OUT/summary.txt

#!/bin/bash
# round 6, session 7: the range look-ahead (fh_bin_prefetch_range): its test, the bench line with it
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s07; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
python3 -c "import sys; sys.path.insert(0,'.'); from frank_amd import _lib as L; print(L.lib.fh_version().decode())" > $OUT/library.txt
timeout 600 python3 -m pytest tests -m gpu -x -q -k "hist or range or cache or prepass or fused" 2>&1 | grep -v "$F" | tail -6 > $OUT/pytest_sel.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
tail -3 $OUT/pytest_sel.txt; tail -3 $OUT/bench_steps20.err
python3 - $OUT/bench_steps20.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print("value", d["value"], "ms/step", d["ms_per_step"])
ex = d["extra"]
for k in ("headline_with_caches", "steady_state", "steady_state_cached", "steady_state_register_resident", "from_host_pipelined", "sweep512_distinct"):
    v = ex.get(k, {})
    print(k, v.get("error") or {a: b for a, b in v.items() if a in ("fits_per_s", "ms_per_step", "ms_per_fit", "upload_bound_fits_per_s")})
PY

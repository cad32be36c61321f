#!/bin/bash
# round 6, session 13: one look at (u, v) for range + histograms: binning tests, the pass alone, the steady state, the bench line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s13; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
timeout 900 python3 -m pytest tests -m gpu -x -q -k "hist or range or cache or prepass or fused or moment or map or bootstrap or sweep or configs1 or pipeline or two_processes" 2>&1 | grep -v "$F" | tail -6 > $OUT/pytest_sel.txt
{ echo "--- the pass alone, nothing remembered: tools/k1_fused.py rows of interest"; timeout 300 python3 tools/k1_fused.py 1e7 300 50 2>&1 | grep -v "$F" | grep "sorted\|fixture" ;
  echo "--- steady state, ring of 4, range cache off, look-ahead"; timeout 300 python3 tools/steady_state.py 2000 --distinct 4 2>&1 | grep -v "$F" | tail -1 | cut -c1-160
  echo "--- steady state, one table, caches on"; timeout 300 python3 tools/steady_state.py 2000 2>&1 | grep -v "$F" | tail -1 | cut -c1-160
  echo "--- steady state, ring of 4, range cache off, look-ahead"; timeout 300 python3 tools/steady_state.py 2000 --distinct 4 2>&1 | grep -v "$F" | tail -1 | cut -c1-160
} > $OUT/one_look.txt 2>&1
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_noextras.json 2> $OUT/bench.err
tail -3 $OUT/pytest_sel.txt; cat $OUT/one_look.txt; python3 - $OUT/bench_noextras.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print("value", d["value"], "binning", {k: d["roofline_binning"][k] for k in ("pass_ms", "frac", "range_kernel_ms", "gram_kernel_ms")})
PY

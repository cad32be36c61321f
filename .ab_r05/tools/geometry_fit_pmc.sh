#!/bin/bash
# HBM traffic of the geometry-fit kernels from the counters (separate --pmc passes, kernel trace only):
#   bash tools/geometry_fit_pmc.sh   (through gpurun, from the repo root)  ->  gpurun_out/r03_geom/pmc_geometry_fit.json
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_geom
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pg_pmc; mkdir -p /tmp/pg_pmc
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pg_pmc/p$i -o p -- python3 $ROOT/tools/geometry_fit_bench.py 1e7 device > /tmp/pg_pmc/p$i.log 2>&1 || echo "group $i failed"
done
cd $ROOT
timeout 100 python3 tools/pmc_summary.py $OUT/all.json /tmp/pg_pmc/p[0-9] > /dev/null
python3 - $OUT <<'PY'
import json, os, sys
out = sys.argv[1]
d = json.load(open(os.path.join(out, "all.json")))
keep = {k: e for k, e in d.items() if any(s in k for s in ("vis_residual", "gauss_", "fd_normal", "fold_"))}
json.dump(keep, open(os.path.join(out, "pmc_geometry_fit.json"), "w"), indent=1)
for k, e in keep.items():
    print(k, e.get("hbm_bytes_per_launch"), e.get("duration_ms_mean_under_pmc"))
PY
rm -f $OUT/all.json

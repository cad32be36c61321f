#!/bin/bash
# Per-kernel times of the binning pass under the environment given (run on the GPU box through gpurun):
#   FRANK_AMD_K1_WPB=8 FRANK_AMD_K1_BLOCKS=512 bash tools/k1_kernels.sh tag
TAG=${1:-default}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/k1k
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
K1_TAG=$TAG timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o s -- python3 $ROOT/tools/k1_pass.py 1e7 300 30 > /tmp/p_$TAG.log 2>&1
grep "tag=" /tmp/p_$TAG.log
f=/tmp/prof_$TAG/s_kernel_stats.csv
if [ -f "$f" ]; then
  cp "$f" $OUT/stats_$TAG.csv
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ("vr_gram", "vr_finish", "uv_hist", "bucket_scan", "deproject_scatter", "piece_moments", "segment_moments", "bucket_factor", "bin_gram2", "reduce_partials", "deproject_kernel", "bucket_")
tot = 0.0
for r in rows:
    if any(k in r["Name"] for k in keep) and int(r["Calls"]) >= 30:
        name = r["Name"].split("(")[1 if r["Name"].startswith("(") else 0] if False else r["Name"]
        short = [k for k in keep if k in r["Name"]][0] + ("<range>" if "Lb0" in r["Name"] or "<false" in r["Name"] else "")
        us = float(r["AverageNs"]) / 1e3
        tot += us
        print("   %-28s %4d calls  %8.1f us" % (r["Name"][:60].replace("(anonymous namespace)::", ""), int(r["Calls"]), us))
print("   sum of the pass's kernels: %.1f us" % tot)
PY
fi

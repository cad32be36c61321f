#!/bin/bash
# Profiles of round 4 on the GPU box (through gpurun from the repo root):  bash tools/profile_r04.sh [tag]
# Writes gpurun_out/<tag>/{kernel_stats.csv, bench_under_rocprof.json, pmc_binning.json, pmc_binning_first_sight.json,
# pmc_fit_loop.json (one CU), pmc_fit_loop_cluster.json, library.txt}; copy to profiles/.
set -u
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG; mkdir -p /tmp/prof_$TAG
python3 -c "import sys; sys.path.insert(0, '$ROOT'); from frank_amd import _lib as L; print(L.lib.fh_version().decode())" > $OUT/library.txt
# 1) per-kernel times of the SAME command the driver runs (minus the secondary workloads)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/stats -o s -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> /tmp/prof_$TAG/stats.log
[ -f /tmp/prof_$TAG/stats/s_kernel_stats.csv ] && cut -d, -f1-8 /tmp/prof_$TAG/stats/s_kernel_stats.csv > $OUT/kernel_stats.csv
# 2) counters, separate passes (kernel trace only): the binning pass of 1e7 visibilities at N = 300, repeated on the same rows
#    (the histogram of the last pass is reused) and with every pass looking at (u, v) again
GROUPS_BASE=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum")
GROUPS_MFMA=("SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES")
run_group () {  # name, env assignment ("" for none), "mfma" or "", then the command
  local name=$1 envkv=$2 extra=$3; shift 3
  local groups=("${GROUPS_BASE[@]}")
  [ "$extra" = "mfma" ] && groups+=("${GROUPS_MFMA[@]}")
  local i=0
  for grp in "${groups[@]}"; do
    i=$((i+1))
    ( [ -n "$envkv" ] && export $envkv; timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/prof_$TAG/${name}_$i -o p -- "$@" > /tmp/prof_$TAG/${name}_$i.log 2>&1 ) || echo "$name group $i failed"
  done
  ( cd $ROOT && timeout 100 python3 tools/pmc_summary.py $OUT/${name}_all.json /tmp/prof_$TAG/${name}_[0-9]* > /dev/null )
}
run_group k1 "" "" python3 $ROOT/tools/k1_pass.py 1e7 300 3
run_group k1first "FRANK_AMD_K1_NO_HIST_CACHE=1" "" python3 $ROOT/tools/k1_pass.py 1e7 300 3
# ... and the fit loop kernel on one N = 300 fit: one compute unit, and a cluster of five workgroups
run_group k2 "FRANK_AMD_K2_CLUSTER=1" mfma python3 $ROOT/tools/k2_quick.py 300
run_group k2cl "" mfma python3 $ROOT/tools/k2_quick.py 300
cd $ROOT
python3 - $OUT <<'PY'
import json, sys, os
out = sys.argv[1]
lib = open(os.path.join(out, "library.txt")).read().strip()
keep1 = ("uv_hist", "bucket_scan", "deproject_scatter", "piece_moments", "bucket_factor2", "vr_gram", "vr_finish")
for src, dst, pred in (("k1_all.json", "pmc_binning.json", lambda k: any(s in k for s in keep1) and "<false>" not in k),
                       ("k1first_all.json", "pmc_binning_first_sight.json", lambda k: any(s in k for s in keep1) and "<false>" not in k),
                       ("k2_all.json", "pmc_fit_loop.json", lambda k: "fit_loop" in k),
                       ("k2cl_all.json", "pmc_fit_loop_cluster.json", lambda k: "fit_loop" in k)):
    try:
        d = json.load(open(os.path.join(out, src)))
    except Exception as e:
        print(src, "missing", e)
        continue
    sel = {k: e for k, e in d.items() if pred(k)}
    sel["_library"] = lib
    json.dump(sel, open(os.path.join(out, dst), "w"), indent=1)
    print(dst, {k: (e.get("hbm_bytes_per_launch"), e.get("duration_ms_mean_under_pmc"), e.get("dispatches")) for k, e in sel.items() if isinstance(e, dict)})
    os.remove(os.path.join(out, src))
PY
head -14 $OUT/kernel_stats.csv

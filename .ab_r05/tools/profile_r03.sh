#!/bin/bash
# Profiles of round 3 on the GPU box (through gpurun from the repo root):  bash tools/profile_r03.sh [tag]
# Writes gpurun_out/<tag>/{kernel_stats.csv, bench_under_rocprof.json, pmc_binning.json, pmc_fit_loop.json}; copy to profiles/.
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG; mkdir -p /tmp/prof_$TAG
# 1) per-kernel times of the SAME command the driver runs (minus the secondary workloads)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/stats -o s -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> /tmp/prof_$TAG/stats.log
[ -f /tmp/prof_$TAG/stats/s_kernel_stats.csv ] && cut -d, -f1-8 /tmp/prof_$TAG/stats/s_kernel_stats.csv > $OUT/kernel_stats.csv
# 2) counters, separate passes (kernel trace only): the binning pass of 1e7 visibilities at N = 300 ...
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/prof_$TAG/k1_$i -o p -- python3 $ROOT/tools/k1_pass.py 1e7 300 3 > /tmp/prof_$TAG/k1_$i.log 2>&1 || echo "binning group $i failed"
done
# ... and the fit loop kernel on one N = 300 fit
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d /tmp/prof_$TAG/k2_$i -o p -- python3 $ROOT/tools/k2_quick.py 300 > /tmp/prof_$TAG/k2_$i.log 2>&1 || echo "fit loop group $i failed"
done
cd $ROOT
timeout 100 python3 tools/pmc_summary.py $OUT/pmc_binning_all.json /tmp/prof_$TAG/k1_[0-9]* > /dev/null
timeout 100 python3 tools/pmc_summary.py $OUT/pmc_fit_loop_all.json /tmp/prof_$TAG/k2_[0-9]* > /dev/null
python3 - $OUT <<'PY'
import json, sys, os
out = sys.argv[1]
keep1 = ("uv_hist", "bucket_scan", "deproject_scatter", "piece_moments", "bucket_factor2", "vr_gram", "vr_finish")
d = json.load(open(os.path.join(out, "pmc_binning_all.json")))
json.dump({k: e for k, e in d.items() if any(s in k for s in keep1) and "<false>" not in k}, open(os.path.join(out, "pmc_binning.json"), "w"), indent=1)
d = json.load(open(os.path.join(out, "pmc_fit_loop_all.json")))
json.dump({k: e for k, e in d.items() if "fit_loop" in k}, open(os.path.join(out, "pmc_fit_loop.json"), "w"), indent=1)
for f in ("pmc_binning.json", "pmc_fit_loop.json"):
    dd = json.load(open(os.path.join(out, f)))
    print(f, {k: (e.get("hbm_bytes_per_launch"), e.get("duration_ms_mean_under_pmc")) for k, e in dd.items()})
PY
rm -f $OUT/pmc_binning_all.json $OUT/pmc_fit_loop_all.json
head -12 $OUT/kernel_stats.csv

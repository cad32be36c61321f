#!/bin/bash
# Profiles of one round on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh r02
# Writes under gpurun_out/<tag>/prof; copy the summaries to profiles/ afterwards.
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1) per-kernel times of the SAME command the driver runs
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
# 2) counters, separate passes (no --stats, no other trace domains)
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/pmc_$name -o p -- python3 $ROOT/tools/k1_quick.py > $OUT/pmc_$name.log 2>&1
done
cd $ROOT
python3 tools/pmc_summary.py $OUT/pmc_hbm.json $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
python3 tools/pmc_summary.py $OUT/pmc_compute.json $OUT/pmc_SQ_* $OUT/pmc_GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py --stats $(find $OUT/stats -name "*_results.db" | head -1) $OUT/kernel_stats.csv  # (times in microseconds)
# keep the merge small: drop the raw traces
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete

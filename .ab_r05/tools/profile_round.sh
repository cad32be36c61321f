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
# 2) counters, separate passes (no --stats, no other trace domains): the fit loop kernel on one N = 300 fit ...
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/k2_$i -o p -- python3 $ROOT/tools/k2_quick.py 300 > $OUT/k2_$i.log 2>&1
done
# ... and the kernels of the binning pass
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/k1_$i -o p -- python3 $ROOT/tools/k1_quick.py > $OUT/k1_$i.log 2>&1
done
cd $ROOT
python3 tools/pmc_summary.py $OUT/pmc_fit_loop.json $OUT/k2_[0-9]* > $OUT/summary.log
python3 tools/pmc_summary.py $OUT/pmc_binning.json $OUT/k1_[0-9]* >> $OUT/summary.log
python3 tools/pmc_summary.py --stats $(find $OUT/stats -name "*_results.db" | head -1) $OUT/kernel_stats.csv >> $OUT/summary.log  # (times in microseconds)
# keep the merge small: drop the raw traces
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*_results.db" -delete

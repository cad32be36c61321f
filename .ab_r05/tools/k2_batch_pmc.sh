#!/bin/bash
# L2 / fabric counters of B concurrent fit loops in one launch:  bash tools/k2_batch_pmc.sh <tag> [B]
set -u
TAG=${1:-r03}
B=${2:-128}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG/prof_k2_batch$B
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_$i -o p -- python3 $ROOT/tools/k2_batch_pmc.py $B > $OUT/pmc_$i.log 2>&1
  tail -1 $OUT/pmc_$i.log
done
cd $ROOT
python3 tools/pmc_summary.py $OUT/pmc.json $OUT/pmc_[0-9]* 2>&1 | tail -5
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*_results.db" -size +8M -delete

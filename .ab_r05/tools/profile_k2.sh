#!/bin/bash
# Counters of fit_loop_kernel on one N = 300 fit (run through gpurun from the repo root):  bash tools/profile_k2.sh r02
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG/prof_k2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_INSTS_MFMA" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/pmc_$i -o p -- python3 $ROOT/tools/k2_quick.py 300 > $OUT/pmc_$i.log 2>&1
done
cd $ROOT
python3 tools/pmc_summary.py $OUT/pmc_k2.json $OUT/pmc_[0-9]*
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*_results.db" -size +8M -delete

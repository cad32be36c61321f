#!/bin/bash
# Counters of the binning pass (separate --pmc passes, kernel trace only):  bash tools/k1_pmc.sh tag   -> gpurun_out/k1pmc_<tag>.json
TAG=${1:-default}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=/tmp/k1pmc_$TAG
rm -rf $OUT; mkdir -p $OUT $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVES" \
           "TCC_REQ_sum TCC_WRITE_sum TCC_READ_sum TCC_ATOMIC_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i -o p -- python3 $ROOT/tools/k1_pass.py 1e7 300 3 > $OUT/g$i.log 2>&1 || echo "group $i ($grp) failed: $(tail -2 $OUT/g$i.log)"
done
cd $ROOT
timeout 100 python3 tools/pmc_summary.py $ROOT/gpurun_out/k1pmc_$TAG.json $OUT/g[0-9]* > /dev/null 2>&1
python3 - $ROOT/gpurun_out/k1pmc_$TAG.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, e in d.items():
    if not any(s in k for s in ("uv_hist", "bucket_scan", "deproject_scatter", "piece_moments", "bucket_factor", "bin_gram2", "reduce_partials")):
        continue
    print(k[:50])
    print("   ", {c: (round(v, 1) if isinstance(v, float) else v) for c, v in e.items() if not c.startswith("launches_")})
PY

#!/bin/bash
# Round 5, first GPU session: evidence for the loaded fit loop.  bash tools/r05_session1.sh   (from the repo root, through gpurun)
#  1. clock + cycles per pass against the number of resident loops (fh_ctx_loop_clocks), and at steady state
#  2. the same with the double-store probe library (2 MB more of stores per pass)
#  3. counters of 256 resident loops: bytes beyond the L2, L2 hit rate, wait fractions
#  4. clusters of 2 / 3 / 4 workgroups as the throughput form: fits/s against outstanding fits
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s1
mkdir -p $OUT
cd $ROOT
export FRANK_AMD_SWEEP_NO_CLUSTERS=1
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.5; done ) > $OUT/smi.log 2>&1 &
SMI=$!
echo "== 1 clocks, shipped library" | tee $OUT/clocks.txt
timeout 600 python3 tools/k2_loaded.py --steady 1 8 32 64 128 192 256 2>&1 | tee -a $OUT/clocks.txt
echo "== 2 clocks, double-store probe" | tee -a $OUT/clocks.txt
FRANK_AMD_LIB=$ROOT/frank_amd/libfrank_hip_probe.so timeout 600 python3 tools/k2_loaded.py --steady 1 64 128 256 2>&1 | tee -a $OUT/clocks.txt
kill $SMI
sort $OUT/smi.log | uniq -c | sort -rn | head -12 > $OUT/smi_summary.txt
echo "== 4 clusters as the throughput form" | tee $OUT/clusters.txt
for spec in "1:8 32 64 120 240" "2:8 32 64 120" "3:8 32 64 80" "4:8 32 60" "6:8 32 40"; do
  g=${spec%%:*}; ns=${spec#*:}
  echo "-- cluster of $g" | tee -a $OUT/clusters.txt
  FRANK_AMD_K2_CLUSTER=$g FRANK_AMD_K2_CLUSTER_FITS=1000 timeout 300 python3 tools/k2_concurrency.py $ns 2>&1 | tee -a $OUT/clusters.txt
done
echo "== 3 counters, 256 loops resident"
bash tools/k2_batch_pmc.sh r05s1 256 2>&1 | tail -12
cp $OUT/prof_k2_batch256/pmc.json $OUT/pmc_fit_loop_256.json 2>/dev/null
ls $OUT

#!/bin/bash
# round 6, session 3: the fused pre-pass under load (steady state of the pipeline) and its bytes (counters)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s03; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
python3 -c "import sys; sys.path.insert(0,'.'); from frank_amd import _lib as L; print(L.lib.fh_version().decode())" > $OUT/library.txt
timeout 600 python3 tools/k1_fused.py 1e7 300 50 2>&1 | grep -v "$F" > $OUT/k1_fused.txt
{ for fz in 0 1; do
    echo "--- FRANK_AMD_K1_FUSED=$fz: tools/steady_state.py 2000"; FRANK_AMD_K1_FUSED=$fz timeout 300 python3 tools/steady_state.py 2000 2>&1 | grep -v "$F" | tail -2
    echo "--- FRANK_AMD_K1_FUSED=$fz: tools/steady_state.py 2000 --distinct 4"; FRANK_AMD_K1_FUSED=$fz timeout 300 python3 tools/steady_state.py 2000 --distinct 4 2>&1 | grep -v "$F" | tail -2
  done; } > $OUT/steady_state_fused.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s03; mkdir -p /tmp/prof_s03
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  ( export FRANK_AMD_K1_FUSED=1; timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/prof_s03/k1f_$i -o p -- python3 $ROOT/tools/k1_pass.py 1e7 300 3 > /tmp/prof_s03/k1f_$i.log 2>&1 ) || echo "group $i failed"
done
( cd $ROOT && timeout 100 python3 tools/pmc_summary.py $OUT/pmc_binning_fused_all.json /tmp/prof_s03/k1f_[0-9]* > /dev/null )
cat $OUT/k1_fused.txt $OUT/steady_state_fused.txt
python3 - $OUT/pmc_binning_fused_all.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, e in d.items():
    if isinstance(e, dict): print(k[:70], e.get("hbm_bytes_per_launch"), e.get("duration_ms_mean_under_pmc"), e.get("dispatches"))
PY

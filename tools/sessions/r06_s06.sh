#!/bin/bash
# round 6, session 6: GPU tests, smoke, the driver's bench command on the new bench.py (headline on the ring, caches off), counters of
# the LogNormal kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s06; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
python3 -c "import sys; sys.path.insert(0,'.'); from frank_amd import _lib as L; print(L.lib.fh_version().decode())" > $OUT/library.txt
timeout 1200 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -12 > $OUT/pytest_gpu.txt
timeout 300 python3 __graft_entry__.py smoke 2>&1 | grep -v "$F" | tail -8 > $OUT/smoke.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s06; mkdir -p /tmp/prof_s06
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/prof_s06/ln_$i -o p -- python3 $ROOT/tools/ln_fullsize.py 1e7 linear > /tmp/prof_s06/ln_$i.log 2>&1 || echo "group $i failed"
done
( cd $ROOT && timeout 100 python3 tools/pmc_summary.py $OUT/pmc_lognormal_all.json /tmp/prof_s06/ln_[0-9]* > /dev/null )
cd $ROOT
tail -4 $OUT/pytest_gpu.txt; tail -3 $OUT/smoke.txt; cut -c1-600 $OUT/bench_steps20.json; tail -5 $OUT/bench_steps20.err
python3 - $OUT/pmc_lognormal_all.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, e in d.items():
    if isinstance(e, dict) and "lognormal" in k: print(k[:70], e)
PY

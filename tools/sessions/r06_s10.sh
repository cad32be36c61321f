#!/bin/bash
# round 6, session 10: GPU tests, smoke, the driver's bench command on the new bench.py (headline on the ring, caches off), counters of
# the LogNormal kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s10; mkdir -p $OUT
F='^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl'
python3 -c "import sys; sys.path.insert(0,'.'); from frank_amd import _lib as L; print(L.lib.fh_version().decode())" > $OUT/library.txt
timeout 1200 python3 -m pytest tests -m gpu -x -q 2>&1 | grep -v "$F" | tail -12 > $OUT/pytest_gpu.txt
timeout 300 python3 __graft_entry__.py smoke 2>&1 | grep -v "$F" | tail -8 > $OUT/smoke.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
cd $ROOT
tail -4 $OUT/pytest_gpu.txt; tail -3 $OUT/smoke.txt; cut -c1-600 $OUT/bench_steps20.json; tail -5 $OUT/bench_steps20.err

#!/bin/bash
# round 6, session 1: the state of the tree as the round starts -- GPU tests, smoke, the driver's bench line
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r06s01; mkdir -p $OUT
python3 -c "import sys; sys.path.insert(0,'.'); from frank_amd import _lib as L; print(L.lib.fh_version().decode())" > $OUT/library.txt
timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $OUT/pytest_gpu.txt
timeout 300 python3 __graft_entry__.py smoke 2>&1 | tail -8 > $OUT/smoke.txt
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
cat $OUT/pytest_gpu.txt; tail -3 $OUT/smoke.txt; cut -c1-400 $OUT/bench_steps20.json

#!/bin/bash
# the round's last session: profile set, bench lines and the GPU suite on the final binary
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05s46; mkdir -p $OUT
bash tools/profile_r05.sh r05final > $OUT/profile.log 2>&1
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
python3 __graft_entry__.py smoke > $OUT/smoke.txt 2>&1

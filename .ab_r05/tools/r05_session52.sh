#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05s52; mkdir -p $OUT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err

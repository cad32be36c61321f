#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05s40; mkdir -p $OUT
python3 bench.py --steps 20 --warmup 5 --no-extras > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
FRANK_AMD_K2_RR=0 python3 bench.py --no-cpu-baseline --no-sharded > $OUT/bench_rr0.json 2> $OUT/bench_rr0.err
python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05s44; mkdir -p $OUT
python -m pytest tests -m gpu -x -q -k "lognormal or bootstrap" > $OUT/pytest_ln.txt 2>&1; tail -3 $OUT/pytest_ln.txt
python3 bench.py --no-cpu-baseline --no-sharded > $OUT/bench.json 2> $OUT/bench.err

#!/bin/bash
# round 6, session 25: the bench lines that go with the profile set (same library, nothing relinked): the driver's command and the default run
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06; mkdir -p $OUT
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
timeout 1200 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cut -c1-400 $OUT/bench_steps20.json; echo; cut -c1-300 $OUT/bench_default.json; tail -3 $OUT/bench_steps20.err $OUT/bench_default.err

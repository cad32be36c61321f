#!/bin/bash
# round 6, session 12: the profile set of the round on the final library (tools/profile_r06.sh) and the bench lines that go with it
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06; mkdir -p $OUT
timeout 2400 bash tools/profile_r06.sh r06 > $OUT/profile_r06.log 2>&1
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> $OUT/bench_steps20.err
timeout 900 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -30 $OUT/profile_r06.log | cut -c1-250; cut -c1-300 $OUT/bench_steps20.json; cut -c1-300 $OUT/bench_default.json; cat $OUT/steady_state_under_rocprof.txt | tail -2 | cut -c1-200

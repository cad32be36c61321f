#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05s30; mkdir -p $OUT
timeout 1500 python3 tools/size_sweep_cluster.py 2>&1 | tail -6 | tee $OUT/size_sweep_cluster.txt
timeout 900 python3 tools/defer_sweep.py 47 319 1 2>&1 | tail -3 | tee $OUT/defer_sweep.txt
timeout 900 python3 tools/hyper_sweep_cluster.py 2>&1 | tail -4 | tee $OUT/hyper_sweep_cluster.txt
bash tools/steady_state_trace_r05.sh 2>&1 | tail -24 | cut -c1-220 | tee $OUT/steady_trace.txt

#!/bin/bash
# the driver's 8-rank command line on the one GPU of the box (HostComm), on the final library
cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/r05s54; mkdir -p $OUT
( time timeout 1700 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 8 --steps 20 --warmup 5 > $OUT/bench8.json 2> $OUT/bench8.err ) 2>&1 | tail -4 | tee $OUT/bench8.txt
echo "exit $?" >> $OUT/bench8.txt

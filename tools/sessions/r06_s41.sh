#!/bin/bash
# round 6, session 41: is the slow 20-step line of a fresh box a cold device?  bench without and with the untimed pre-warm, twice each
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
OUT=$ROOT/gpurun_out/r06s41; mkdir -p $OUT
for tag in cold0 warm2 cold0b warm2b; do
  case $tag in cold0*) P=0;; *) P=2;; esac
  timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --prewarm-seconds $P > $OUT/$tag.json 2> $OUT/$tag.err
  python3 -c "
import json,sys
d=json.loads(open('$OUT/$tag.json').read()); print('$tag', 'prewarm', $P, 'value', round(d['value'],1))"
done

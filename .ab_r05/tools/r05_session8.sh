#!/bin/bash
# Round 5, session 8: a floor of compute units under the binning stream; where map_visibilities spends its time; the 8-rank bench
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s8
mkdir -p $OUT
cd $ROOT
run () { python3 tools/steady_state.py 2>&1 | tail -1 | sed -e "s/.*'fits_per_s': //" -e "s/, 'steps.*//" ; }
{
for pair in 0 1; do
for spec in "0 240" "32 240" "48 240" "64 240" "64 320" "96 240" "96 320" "128 320"; do
  set -- $spec
  echo -n "pair=$pair reserve=$1 slots=$2: "
  FRANK_AMD_K2_PAIR=$pair FRANK_AMD_FIT_RESERVE_CUS=$1 FRANK_AMD_FIT_SLOTS=$2 run
done
done
for b in 32 64 96; do echo -n "partition BIN_CUS=$b pair=0: "; FRANK_AMD_K2_PAIR=0 BIN_CUS=$b run; done
} 2>&1 | tee $OUT/reserve.txt
timeout 300 python3 tools/map_phases.py 2>&1 | tail -4 | tee $OUT/map_phases.txt
echo "== 8 ranks on this one GPU (HostComm), the driver's command line" | tee $OUT/bench8.txt
( time timeout 1700 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 8 --steps 20 --warmup 5 > $OUT/bench8.json 2> $OUT/bench8.err ) 2>&1 | tail -4 | tee -a $OUT/bench8.txt
echo "exit code $?" | tee -a $OUT/bench8.txt
tail -c 600 $OUT/bench8.err
python3 - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r05s8/bench8.json").read().strip().split("\n")[-1])
    print("value", d["value"], "n_gpus", d["n_gpus"], "sharded", {k: d["sharded_fit"].get(k) for k in ("comm", "s_per_fit", "nvis_per_rank", "iterations", "error")},
          "sweep", {k: d["sweep512_multi"].get(k) for k in ("fits_per_s", "failed", "error")})
except Exception as e:
    print("no line:", e)
PY

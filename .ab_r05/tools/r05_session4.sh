#!/bin/bash
# Round 5, session 4: what bounds the steady state -- the binning stream or the loops?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s4
mkdir -p $OUT
cd $ROOT
{
echo "== steady state against the size of the table (pair library)"
for n in 1e5 1e6 3e6 1e7; do NVIS=$n timeout 300 python3 tools/steady_state.py 2>&1 | tail -1; done
echo "== the same, no-pair library"
for n in 1e5 1e7; do FRANK_AMD_LIB=$ROOT/frank_amd/libfrank_hip_nopair.so NVIS=$n timeout 300 python3 tools/steady_state.py 2>&1 | tail -1; done
echo "== the same, kernel of rounds 2-4"
for n in 1e5 1e7; do FRANK_AMD_K2_DEFER=0 NVIS=$n timeout 300 python3 tools/steady_state.py 2>&1 | tail -1; done
echo "== fits per launch / streams (pair library, 1e7 rows)"
for spec in "32 6" "32 8" "48 6" "64 8" "40 6"; do set -- $spec; echo "batch $1 streams $2"; FRANK_AMD_FIT_BATCH=$1 FRANK_AMD_FIT_STREAMS=$2 timeout 300 python3 tools/steady_state.py 2>&1 | tail -1; done
echo "== binning stream alone: steps per second without fits"
timeout 300 python3 - <<'PY'
import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import bench
from frank_amd import _lib as L
from frank_amd.mock import mock_disc_visibilities
f = bench.Fitter(L, 300, 0)
f.nfit = 10_000_000
f.upload(*mock_disc_visibilities(f.nfit, seed=0, noise_seed=50))
f.fit()
for rep in range(2):
    f.sync(); t0 = time.perf_counter()
    for i in range(400):
        f.bin()
        L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 0, None, None, None, None, None))
    f.sync(); dt = time.perf_counter() - t0
print("bin + finalize alone: %.3f ms per step" % (1e3 * dt / 400))
PY
} 2>&1 | tee $OUT/steady.txt

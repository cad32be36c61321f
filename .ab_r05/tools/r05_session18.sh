#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05s18
mkdir -p $OUT
cd $ROOT
timeout 1800 python3 -m pytest tests -m gpu -x -q --timeout 1500 -k "lognormal or staged or sweep" 2>&1 | tail -6 | tee $OUT/tests.txt
python3 - <<'PY' 2>&1 | tee $OUT/ln_batched.txt
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
from frank_amd import _lib as L
from frank_amd.mock import mock_disc_visibilities
N = 300
f = bench.Fitter(L, N, 0)
f.nfit = 1_000_000
f.upload(*mock_disc_visibilities(10_000_000, seed=0, noise_seed=50))
f.fit()
f.bin(1_000_000)
L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 0, None, None, None, None, None))
al, ws = np.meshgrid(np.linspace(1.05, 1.5, 8), np.logspace(-4, -1, 8))
al, ws = np.ascontiguousarray(al.ravel()), np.ascontiguousarray(ws.ravel())
B = al.size
p0 = np.full(B, 1e-35)
for env in ("0", "1"):
    os.environ["FRANK_AMD_LN_SWEEP_CLUSTERS"] = env
    s_map, p = np.empty((B, N)), np.empty((B, N))
    nit, st = (ctypes.c_int * B)(), (ctypes.c_int * B)()
    stats = np.zeros(9 * B, dtype=np.int64)
    best = None
    for rep in range(2):
        t0 = time.perf_counter()
        L.check(L.lib.fh_fit_lognormal_batched(f.ctx, None, None, B, L.ptr(al), L.ptr(p0), L.ptr(ws), 1e-3, 2000, 1e10, L.ptr(s_map), L.ptr(p), nit, st,
                                               stats.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    its = np.sort(np.array(list(nit)))[::-1]
    print("LN_SWEEP_CLUSTERS=%s: %.1f fits/s (%.2f s); passes, longest first: %s; sha %s" % (
        env, B / best, best, its[:8].tolist(), __import__("hashlib").sha256(s_map.tobytes() + p.tobytes()).hexdigest()[:12]), flush=True)
PY

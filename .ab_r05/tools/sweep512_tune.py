"""BASELINE configs[4] on one GPU (512 fits of one 1e6-visibility mapping, bench.py extra.sweep512) under the switches of the
batched launch: how many of the longest points go to clusters (FRANK_AMD_SWEEP_CLUSTERS), which form of the one-workgroup kernel
(FRANK_AMD_K2_PAIR / FRANK_AMD_K2_DEFER).   python3 tools/sweep512_tune.py   (prints fits/s and the iteration histogram once)
"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from frank_amd import _lib as L  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

N = bench.N_COLL
f = bench.Fitter(L, N, 0)
f.nfit = 1_000_000
# BENCH_TABLE=1: the bench's own workload -- the first 1e6 rows of the 1e7-row headline table (another draw than 1e6 rows of seed 0)
nrows = 10_000_000 if os.environ.get("BENCH_TABLE") else f.nfit
f.upload(*mock_disc_visibilities(nrows, seed=0, noise_seed=50))
h = bench.HYPER
f.fit()
al, ws = np.meshgrid(np.linspace(1.01, 1.5, 32), np.logspace(-4, -1, 16))
al, ws = np.ascontiguousarray(al.ravel()), np.ascontiguousarray(ws.ravel())
B = al.size
p0 = np.full(B, h["p0"])
mu, pp = np.empty((B, N)), np.empty((B, N))
niter, status = (ctypes.c_int * B)(), (ctypes.c_int * B)()
best = None
for rep in range(3):
    f.sync()
    t0 = time.perf_counter()
    f.bin()
    L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 0, None, None, None, None, None))
    L.check(L.lib.fh_fit_normal_batched(f.ctx, None, None, B, L.ptr(al), L.ptr(p0), L.ptr(ws), h["tol"], h["max_iter"],
                                        L.ptr(mu), L.ptr(pp), niter, status))
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
its = np.array(list(niter))
print("%s: %.0f fits/s (best of 3: %.3f s); passes: total %d, max %d, %d at max_iter; sha %s" % (
    " ".join("%s=%s" % (k[10:], v) for k, v in sorted(os.environ.items()) if k.startswith("FRANK_AMD_")) or "defaults",
    B / best, best, int(its.sum()) + 2 * B, int(its.max()), int((its >= h["max_iter"]).sum()),
    __import__("hashlib").sha256(mu.tobytes() + pp.tobytes()).hexdigest()[:12]), flush=True)
if os.environ.get("PRINT_ITS"):
    print("passes per fit, descending:", " ".join(str(int(x) + 2) for x in np.sort(its)[::-1]))

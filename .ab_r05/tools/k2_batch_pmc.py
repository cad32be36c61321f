"""B identical fits in ONE launch of the fit loop (fh_fit_normal_batched: one workgroup per fit, all resident together): what B
concurrent loops cost each other.  One kernel, so it can run under `rocprofv3 --pmc` (the pipeline's launches would serialise).
    python3 tools/k2_batch_pmc.py [B=128] [N=300]
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else bench.N_COLL
f = bench.Fitter(L, N, 0)
f.nfit = 1_000_000
f.upload(*mock_disc_visibilities(f.nfit, seed=0, noise_seed=50))
h = bench.HYPER
f.bin()
L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 0, None, None, None, None, None))
al, p0, ws = np.full(B, h["alpha"]), np.full(B, h["p0"]), np.full(B, h["wsmooth"])
mu, pp = np.empty((B, N)), np.empty((B, N))
niter, status = (ctypes.c_int * B)(), (ctypes.c_int * B)()
for rep in range(2):
    f.sync()
    t0 = time.perf_counter()
    L.check(L.lib.fh_fit_normal_batched(f.ctx, None, None, B, L.ptr(al), L.ptr(p0), L.ptr(ws), h["tol"], h["max_iter"],
                                        L.ptr(mu), L.ptr(pp), niter, status))
    dt = time.perf_counter() - t0
print("%d fits in one launch: %.1f ms, %d passes each, %.1f us per pass, %.0f fits/s" % (
    B, 1e3 * dt, niter[0] + 2, 1e6 * dt / (niter[0] + 2), B / dt))

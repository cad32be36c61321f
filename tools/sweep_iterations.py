"""Iteration counts over the (alpha, w_smooth) grid of BASELINE configs[4] (512 fits of one 1e6-visibility mapping): which points
are the long ones, i.e. in which order a work queue should hand them out.   python3 tools/sweep_iterations.py"""
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
f.upload(*mock_disc_visibilities(f.nfit, seed=0, noise_seed=50))
h = bench.HYPER
al, ws = np.meshgrid(np.linspace(1.01, 1.5, 32), np.logspace(-4, -1, 16))
al, ws = np.ascontiguousarray(al.ravel()), np.ascontiguousarray(ws.ravel())
B = al.size
p0 = np.full(B, h["p0"])
mu, pp = np.empty((B, N)), np.empty((B, N))
niter, status = (ctypes.c_int * B)(), (ctypes.c_int * B)()
f.bin()
L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 0, None, None, None, None, None))
for rep in range(2):
    t0 = time.perf_counter()
    L.check(L.lib.fh_fit_normal_batched(f.ctx, None, None, B, L.ptr(al), L.ptr(p0), L.ptr(ws), h["tol"], h["max_iter"],
                                        L.ptr(mu), L.ptr(pp), niter, status))
    dt = time.perf_counter() - t0
its = np.array(list(niter)).reshape(16, 32)
print("%.3f s, %d fits/s; sum of iterations %d -> %.3f s of one-CU time at 140 us; longest %d" % (
    dt, B / dt, its.sum(), its.sum() * 140e-6, its.max()))
np.set_printoptions(linewidth=250)
print("rows: w_smooth 1e-4 .. 1e-1, columns: alpha 1.01 .. 1.5")
print(its)

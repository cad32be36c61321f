"""Iteration counts over the (alpha, w_smooth) grid of BASELINE configs[4] (which points run long?).   python tools/sweep_iterations.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from frank_amd import _lib as L  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

N = 300
f = bench.Fitter(L, N, 0)
f.nfit = 1_000_000
f.upload(*mock_disc_visibilities(f.nfit, seed=0, noise_seed=50))
f.bin()
L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 0, None, None, None, None, None))
al, ws = np.meshgrid(np.linspace(1.01, 1.5, 32), np.logspace(-4, -1, 16))
al, ws = np.ascontiguousarray(al.ravel()), np.ascontiguousarray(ws.ravel())
B = al.size
p0 = np.full(B, 1e-15)
mu, pp = np.empty((B, N)), np.empty((B, N))
niter = (ctypes.c_int * B)()
status = (ctypes.c_int * B)()
L.check(L.lib.fh_fit_normal_batched(f.ctx, None, None, B, L.ptr(al), L.ptr(p0), L.ptr(ws), 1e-3, 2000, L.ptr(mu), L.ptr(pp), niter, status))
it = np.array(list(niter)).reshape(16, 32)
np.set_printoptions(linewidth=250)
print("rows: w_smooth 1e-4 .. 1e-1; columns: alpha 1.01 .. 1.5 (first 8)")
print(it[:, :8])
print("total iterations", it.sum(), "of which alpha = 1.01:", it[:, 0].sum())

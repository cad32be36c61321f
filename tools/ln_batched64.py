"""The bench's 64-point LogNormal sweep (8 alpha x 8 w_smooth over one 1e6-visibility mapping, N = 300): staged (default) against the
single launch (FRANK_AMD_LN_CLUSTER=1): seconds, counts, bits.   python3 tools/ln_batched64.py"""
import ctypes
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from frank_amd import _lib as L  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

N = 300
f = bench.Fitter(L, N, 0)
f.nfit = 1_000_000
f.upload(*mock_disc_visibilities(f.nfit, seed=0, noise_seed=50))
al, ws = np.meshgrid(np.linspace(1.2, 1.5, 8), np.logspace(-3, -1, 8))
al, ws = np.ascontiguousarray(al.ravel()), np.ascontiguousarray(ws.ravel())
B = al.size
p0 = np.full(B, 1e-35)
H0, a, b = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
modes = ["staged", "single launch", "staged"] + ["staged, %s left" % x for x in sys.argv[1:]]  # (the development build reads FRANK_AMD_LN_STAGE_LEFT)
for mode in modes:
    os.environ.pop("FRANK_AMD_LN_CLUSTER", None)
    os.environ.pop("FRANK_AMD_LN_STAGE_LEFT", None)
    if mode == "single launch":
        os.environ["FRANK_AMD_LN_CLUSTER"] = "1"
    elif "left" in mode:
        os.environ["FRANK_AMD_LN_STAGE_LEFT"] = mode.split()[1]
    s_map, pp = np.empty((B, N)), np.empty((B, N))
    nit, status = (ctypes.c_int * B)(), (ctypes.c_int * B)()
    f.bin()
    L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 1, None, None, ctypes.byref(H0), ctypes.byref(a), ctypes.byref(b)))
    t0 = time.perf_counter()
    L.check(L.lib.fh_fit_lognormal_batched(f.ctx, None, None, B, L.ptr(al), L.ptr(p0), L.ptr(ws), 1e-3, 2000, 1e5, L.ptr(s_map), L.ptr(pp),
                                           nit, status, None))
    dt = time.perf_counter() - t0
    its = np.array(list(nit))
    print("%-14s %.3f s = %.1f fits/s; iterations min/median/max %d/%d/%d, at max_iter: %d, failed %d; sha %s" % (
        mode, dt, B / dt, its.min(), np.median(its), its.max(), int((its >= 2000).sum()), int(np.sum(np.array(list(status)) != 0)),
        hashlib.sha1(s_map.tobytes() + pp.tobytes() + its.tobytes()).hexdigest()[:12]), flush=True)

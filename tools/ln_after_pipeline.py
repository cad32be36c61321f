"""configs[2] on a context that has run the Normal pipeline first (as bench.py's context has): seconds of three LogNormal fits in a row.
    python3 tools/ln_after_pipeline.py"""
import ctypes
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
f.upload(*mock_disc_visibilities(10_000_000, seed=0, noise_seed=50))
f.fit()
if "--no-pipeline" not in sys.argv:
    f.run_steps(20)
    f.sync()
    bench.steady_state(f, L, 600, 0)
if "--from-host" in sys.argv:  # (bench.py's from_host_pipelined leg: tables uploaded, binned, fitted, destroyed)
    u, v, V, w = mock_disc_visibilities(10_000_000, seed=0, noise_seed=50)
    Vre, Vim = np.ascontiguousarray(V.real), np.ascontiguousarray(V.imag)
    L.check(L.lib.fh_ctx_set_range_cache(f.ctx, 0))
    pend, alive = [], []
    for i in range(60):
        vis = ctypes.c_void_p()
        L.check(L.lib.fh_vis_upload(0, L.ptr(u), L.ptr(v), L.ptr(Vre), L.ptr(Vim), L.ptr(w), w.size, u.size, ctypes.byref(vis)))
        alive.append(vis)
        pend.append(f.submit(vis))
        if (i + 1) % 8 == 0:
            L.check(L.lib.fh_fit_flush(f.ctx))
        while len(alive) > 24:
            f.collect(pend.pop(0))
            L.lib.fh_vis_destroy(alive.pop(0))
    L.check(L.lib.fh_fit_flush(f.ctx))
    while pend:
        f.collect(pend.pop(0))
    f.sync()
    for vis in alive:
        L.lib.fh_vis_destroy(vis)
    L.check(L.lib.fh_ctx_set_range_cache(f.ctx, 1))
h = bench.HYPER
for rep in range(3):
    s_map, p = np.empty(N), np.empty(N)
    nit = ctypes.c_int(0)
    stats = (ctypes.c_int64 * 9)()
    f.bin()
    H0, qmn, qmx = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 1, None, None, ctypes.byref(H0), ctypes.byref(qmn), ctypes.byref(qmx)))
    t1 = time.perf_counter()
    L.check(L.lib.fh_fit_lognormal(f.ctx, None, None, 1.3, 1e-35, 1e-2, h["tol"], h["max_iter"], 1e5, L.ptr(s_map), L.ptr(p), ctypes.byref(nit), None, stats, None, None))
    print("LogNormal fit %d: %.3f s (%d passes, %d Hessians)" % (rep, time.perf_counter() - t1, nit.value, stats[3]), flush=True)

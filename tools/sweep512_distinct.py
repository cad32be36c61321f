"""BASELINE configs[4], distinct-datasets variant, as bench.py's extra.sweep512_distinct runs it: 512 fits over the alpha x w_smooth
grid (32 x 16), every fit binning its OWN 1e6-visibility table (8 resident tables with seeds 0..7 in turn, range cache off, the
look at (u, v) of the next table one step ahead), through the pipeline; the grid goes in order of increasing alpha (longest fits
first).  Two schedules, one process, one SHA over all results: the launches as the pipeline forms them (64 fits each, one compute
unit per fit), and with the first 32 fits -- the longest of the grid -- flushed at once, which puts them on clusters of workgroups
(at most 32 fits outstanding: capi_fit.hip) beside the launches that follow.
    python3 tools/sweep512_distinct.py [reps]"""
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

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
NT, NV, N = 8, 1_000_000, 300
h = bench.HYPER
f = bench.Fitter(L, N, 0)
f.nfit = NV
tabs = []
for sd in range(NT):
    f.upload(*mock_disc_visibilities(NV, seed=sd, noise_seed=50 + sd))
    tabs.append(f.vis)
L.check(L.lib.fh_ctx_set_range_cache(f.ctx, 0))
grid = [(float(x), float(y)) for x in np.linspace(1.01, 1.5, 32) for y in np.logspace(-4, -1, 16)]
slots = L.lib.fh_fit_slots()


def run_grid(points, first_on_clusters=0):
    pend, out = [], []

    def collect(t):
        its = f.collect(t)
        out.append((its, f.mu.copy(), f.p.copy()))
    for i, (ga, gw) in enumerate(points):
        if len(pend) == slots:
            collect(pend.pop(0))
        if i + 1 < len(points):
            L.check(L.lib.fh_bin_prefetch_range(f.ctx, ctypes.byref(f.geom), tabs[(i + 1) % NT], 0, NV))
        f.bin(vis=tabs[i % NT])
        L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 0, None, None, None, None, None))
        t = ctypes.c_int(-1)
        L.check(L.lib.fh_fit_submit(f.ctx, ga, h["p0"], gw, h["tol"], h["max_iter"], ctypes.byref(t)))
        pend.append(t.value)
        if i + 1 == first_on_clusters:
            L.check(L.lib.fh_fit_flush(f.ctx))
    L.check(L.lib.fh_fit_flush(f.ctx))
    for t in pend:
        collect(t)
    return out


for tag, first in (("launches of 64 as they fill", 0), ("the first 32 fits on clusters", 32), ("the first 16 fits on clusters", 16),
                   ("launches of 64 as they fill", 0), ("the first 32 fits on clusters", 32)):
    run_grid(grid[:32])
    f.sync()
    for _ in range(reps):
        t0 = time.perf_counter()
        res = run_grid(grid, first)
        f.sync()
        dt = time.perf_counter() - t0
        its = np.array([r[0] for r in res])
        sha = hashlib.sha1(b"".join(r[1].tobytes() + r[2].tobytes() for r in res) + its.tobytes()).hexdigest()[:12]
        wg, fb = ctypes.c_int(0), ctypes.c_int64(0)
        L.check(L.lib.fh_fit_cluster_info(f.ctx, ctypes.byref(wg), ctypes.byref(fb)))
        print("%-32s %6.0f fits/s (%.3f s); passes min / median / max %d / %d / %d; cluster fall-backs so far %d; sha %s" % (
            tag, len(grid) / dt, dt, its.min(), np.median(its), its.max(), fb.value, sha), flush=True)

"""One pipeline or two?  The fits of a pipeline pass one after the other through ONE context's binning stream (the binning pass, the
finalisation, the q-space operands: ~17 kernels per fit); two contexts on the same device -- their steps taken in turn by one host
thread -- run two such chains beside each other.  The bench's 20-step region and the steady state (ring of 4 table objects, range
cache off, look-ahead), and the distinct-datasets sweep of configs[4], with one context and with two.
    python3 tools/two_contexts.py [steady_steps]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from frank_amd import _lib as L  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

STEADY = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
N = 300
h = bench.HYPER


def make(nctx, nfit):
    fs = [bench.Fitter(L, N, 0) for _ in range(nctx)]
    for f in fs:
        f.nfit = nfit
        L.check(L.lib.fh_ctx_set_range_cache(f.ctx, 0))
    return fs


def run(fs, tabs, k, nfit, hyper=None):
    """k steps, step i on context i % len(fs) and table i % len(tabs); returns the iteration counts in step order"""
    nc = len(fs)
    slots = L.lib.fh_fit_slots() if nc == 1 else max(32, L.lib.fh_fit_slots() // nc)
    pend = [[] for _ in fs]
    its = {}

    def look(i):
        if i < k:
            f = fs[i % nc]
            L.check(L.lib.fh_bin_prefetch_range(f.ctx, ctypes.byref(f.geom), tabs[i % len(tabs)], 0, nfit))
    for i in range(min(nc, k)):
        look(i)
    for i in range(k):
        f, q = fs[i % nc], pend[i % nc]
        if len(q) == slots:
            j, t = q.pop(0)
            its[j] = f.collect(t)
        look(i + nc)
        f.bin(vis=tabs[i % len(tabs)])
        L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 0, None, None, None, None, None))
        a, ws = hyper[i] if hyper else (h["alpha"], h["wsmooth"])
        t = ctypes.c_int(-1)
        L.check(L.lib.fh_fit_submit(f.ctx, a, h["p0"], ws, h["tol"], h["max_iter"], ctypes.byref(t)))
        q.append((i, t.value))
    for f in fs:
        L.check(L.lib.fh_fit_flush(f.ctx))
    for f, q in zip(fs, pend):
        for j, t in q:
            its[j] = f.collect(t)
    for f in fs:
        f.sync()
    return [its[i] for i in range(k)]


# ---- the headline's tables: four table objects of the reference's rows
arrs = mock_disc_visibilities(10_000_000, seed=0, noise_seed=50)
f0 = bench.Fitter(L, N, 0)
tabs = []
for _ in range(4):
    f0.upload(*arrs)
    tabs.append(f0.vis)
del arrs
for nc in (1, 2, 1, 2):
    fs = make(nc, 10_000_000)
    run(fs, tabs, 8, 10_000_000)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        its = run(fs, tabs, 20, 10_000_000)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    t0 = time.perf_counter()
    its = run(fs, tabs, STEADY, 10_000_000)
    dts = time.perf_counter() - t0
    print("1e7-row tables, %d context%s: 20 steps %.1f fits/s (%.1f ms); %d steps %.0f fits/s; passes %d..%d" % (
        nc, "s" if nc > 1 else " ", 20 / best, 1e3 * best, STEADY, STEADY / dts, min(its), max(its)), flush=True)
    del fs
for t in tabs:
    L.lib.fh_vis_destroy(t)
# ---- configs[4], distinct datasets: eight 1e6-row tables in turn, the 32 x 16 grid
tabs = []
for sd in range(8):
    f0.upload(*mock_disc_visibilities(1_000_000, seed=sd, noise_seed=50 + sd))
    tabs.append(f0.vis)
grid = [(float(x), float(y)) for x in np.linspace(1.01, 1.5, 32) for y in np.logspace(-4, -1, 16)]
for nc in (1, 2, 3, 1, 2):
    fs = make(nc, 1_000_000)
    run(fs, tabs, 32, 1_000_000, grid)
    for _ in range(2):
        t0 = time.perf_counter()
        its = run(fs, tabs, len(grid), 1_000_000, grid)
        dt = time.perf_counter() - t0
        print("512 fits over eight 1e6-row tables, %d context%s: %.0f fits/s (%.3f s); passes %d / %d / %d" % (
            nc, "s" if nc > 1 else " ", len(grid) / dt, dt, min(its), int(np.median(its)), max(its)), flush=True)
    del fs

"""B identical fits in ONE launch of the one-workgroup fit loop, all resident, with the clock probe on (fh_ctx_loop_clocks): what a
pass costs in TIME and in shader-clock CYCLES as the device fills -- a pass that is slower in time only is the clock of a loaded
device, one that is slower in cycles is the memory system.  Then (--steady) the pipeline at steady state with the same probe.
    FRANK_AMD_SWEEP_NO_CLUSTERS=1 python3 tools/k2_loaded.py [--steady] [B ...]
"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("FRANK_AMD_SWEEP_NO_CLUSTERS", "1")
import bench  # noqa: E402
from frank_amd import _lib as L  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("-")]
Bs = [int(a) for a in args] or [1, 8, 32, 64, 128, 192, 256]
N = bench.N_COLL
f = bench.Fitter(L, N, 0)
f.nfit = int(float(os.environ.get("NVIS", "1e6")))
f.upload(*mock_disc_visibilities(f.nfit, seed=0, noise_seed=50))
h = bench.HYPER
f.bin()
L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 0, None, None, None, None, None))
out3 = (ctypes.c_int64 * 3)()
L.check(L.lib.fh_ctx_loop_clocks(f.ctx, 1, out3))


def clocks():
    L.check(L.lib.fh_ctx_loop_clocks(f.ctx, 1, out3))
    cyc, ticks, passes = out3[0], out3[1], out3[2]
    if not ticks or not passes:
        return "no fits counted"
    return "clock %.0f MHz, %.1f us and %.0f cycles per pass on the device (%d passes)" % (
        100.0 * cyc / ticks, ticks / 100.0 / passes, cyc / passes, passes)


for B in Bs:
    al, p0, ws = np.full(B, h["alpha"]), np.full(B, h["p0"]), np.full(B, h["wsmooth"])
    mu, pp = np.empty((B, N)), np.empty((B, N))
    niter, status = (ctypes.c_int * B)(), (ctypes.c_int * B)()
    for rep in range(2):
        f.sync()
        clocks()
        t0 = time.perf_counter()
        L.check(L.lib.fh_fit_normal_batched(f.ctx, None, None, B, L.ptr(al), L.ptr(p0), L.ptr(ws), h["tol"], h["max_iter"],
                                            L.ptr(mu), L.ptr(pp), niter, status))
        dt = time.perf_counter() - t0
    print("%4d loops resident: %.1f ms, %.1f us per pass by the host's clock, %.0f fits/s; %s" % (
        B, 1e3 * dt, 1e6 * dt / (niter[0] + 2), B / dt, clocks()), flush=True)

if "--steady" in sys.argv:
    f2 = bench.Fitter(L, N, 0)
    f2.nfit = 10_000_000
    f2.upload(*mock_disc_visibilities(f2.nfit, seed=0, noise_seed=50))
    f2.fit()
    o2 = (ctypes.c_int64 * 3)()
    L.check(L.lib.fh_ctx_loop_clocks(f2.ctx, 1, o2))
    r = bench.steady_state(f2, L, 0, 0)
    L.check(L.lib.fh_ctx_loop_clocks(f2.ctx, 1, o2))
    print("steady state: %.0f fits/s; clock %.0f MHz, %.1f us and %.0f cycles per pass on the device (%d passes)" % (
        r["fits_per_s"], 100.0 * o2[0] / max(o2[1], 1), o2[1] / 100.0 / max(o2[2], 1), o2[0] / max(o2[2], 1), o2[2]), flush=True)

"""Cluster ("latency") mode of the fit loop against the one-workgroup kernel: same bits, time per pass.
   python tools/k2_cluster.py [sizes...]        (FRANK_AMD_K2_CLUSTER / FRANK_AMD_K2_CL_WORKERS are switched in-process)"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
from frank_amd import _lib as L  # noqa: E402
from frank_amd.constants import rad_to_arcsec  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def fit(ctx, N, M, j, alpha, ws, reps=3):
    mu, p, nit = np.empty(N), np.empty(N), ctypes.c_int(0)
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        L.check(L.lib.fh_fit_normal(ctx, L.ptr(M), L.ptr(j), alpha, 1e-15, ws, 1e-3, 2000, L.ptr(mu), L.ptr(p),
                                    ctypes.byref(nit), None, None))
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    ms = ctypes.c_float(0)
    L.check(L.lib.fh_fit_last_kernel_ms(ctx, ctypes.byref(ms)))
    wg, fb = ctypes.c_int(0), ctypes.c_int64(0)
    L.check(L.lib.fh_fit_cluster_info(ctx, ctypes.byref(wg), ctypes.byref(fb)))
    return mu.copy(), p.copy(), nit.value, best, ms.value, wg.value, fb.value


def problem(N):
    """M, j of a fixture when there is one at this size, else a synthetic SPD system of the same scaling."""
    for name in ("fit_N300_1e7.npz", "fit_N100_1e5.npz"):
        g = np.load(os.path.join(GOLD, name))
        if int(g["N"]) == N:
            return np.ascontiguousarray(g["M"]), np.ascontiguousarray(g["j"]), int(g["niter"]), float(g["alpha"]), float(g["wsmooth"])
    from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities
    from frank_amd import FrankFitter, FixedGeometry
    u, v, V, w = mock_disc_visibilities(200000, seed=3, noise_seed=4)
    FF = FrankFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), verbose=False)
    m = FF.preprocess_visibilities(u, v, V, w)
    return np.ascontiguousarray(m["M"]), np.ascontiguousarray(m["j"]), None, 1.05, 1e-4


sizes = [int(a) for a in sys.argv[1:]] or [300]
CLUSTERS = [int(x) for x in os.environ.get("CLUSTERS", "2,3,4,5").split(",")]
WORKERS = [11]
for N in sizes:
    M, j, ref_it, alpha, ws = problem(N)
    dht, ctx = ctypes.c_void_p(), ctypes.c_void_p()
    L.check(L.lib.fh_dht_create(2.0 / rad_to_arcsec, N, 0, ctypes.byref(dht)))
    L.check(L.lib.fh_ctx_create(dht, 0, ctypes.byref(ctx)))
    os.environ["FRANK_AMD_K2_CLUSTER"] = "1"
    os.environ["FRANK_AMD_K2_LL"] = "0"
    mu0, p0, n0, t0, k0, wg0, _ = fit(ctx, N, M, j, alpha, ws)
    print("N=%d  one workgroup, right-looking: %d iterations (reference %s)  %.2f ms  kernel %.2f ms  %.1f us/pass" % (
        N, n0, ref_it, 1e3 * t0, k0, 1e3 * k0 / (n0 + 2)), flush=True)
    os.environ["FRANK_AMD_K2_LL"] = "1"
    mu, p, n, t, k, wg, fb = fit(ctx, N, M, j, alpha, ws)
    print("N=%d  one workgroup, left-looking:  %d iterations  %.2f ms  kernel %.2f ms  %.1f us/pass  bitwise equal: %s  max|dmu|/max %.2e" % (
        N, n, 1e3 * t, k, 1e3 * k / (n + 2), bool(n == n0 and np.array_equal(mu, mu0) and np.array_equal(p, p0)),
        np.abs(mu - mu0).max() / np.abs(mu0).max()), flush=True)
    for g in CLUSTERS:
        for workers in WORKERS:
            os.environ["FRANK_AMD_K2_CLUSTER"] = str(g)
            os.environ["FRANK_AMD_K2_CL_WORKERS"] = str(workers)
            mu, p, n, t, k, wg, fb = fit(ctx, N, M, j, alpha, ws)
            same = bool(n == n0 and np.array_equal(mu, mu0) and np.array_equal(p, p0))
            print("N=%d  cluster %d (ran on %d, fallbacks %d) workers %d: %d iterations  %.2f ms  kernel %.2f ms  %.1f us/pass  bitwise equal: %s  max|dmu|/max %.2e" % (
                N, g, wg, fb, workers, n, 1e3 * t, k, 1e3 * k / (n + 2), same, np.abs(mu - mu0).max() / np.abs(mu0).max()), flush=True)
    L.lib.fh_ctx_destroy(ctx)
    L.lib.fh_dht_destroy(dht)

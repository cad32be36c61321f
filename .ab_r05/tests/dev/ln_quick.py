#!/usr/bin/env python3
"""Quick GPU check of the LogNormal kernel against the CPU oracle (development tool, not a test)."""
import ctypes
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from frank_amd import _lib
from frank_amd.constants import rad_to_arcsec
from frank_amd.hankel import DiscreteHankelTransform
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities
from oracle import oracle as O


def run(N, n, alpha, ws, max_iter=2000, full=True):
    u, v, V, w = mock_disc_visibilities(n, seed=5, noise_seed=6)
    Rmax = 2.0 / rad_to_arcsec
    geom = (MOCK_GEOMETRY["inc"], MOCK_GEOMETRY["PA"], MOCK_GEOMETRY["dRA"], MOCK_GEOMETRY["dDec"])
    m = O.map_visibilities(N, Rmax, geom, u, v, V, w, check_qbounds=False)
    M, j = m["M"], m["j"]
    D = O.DHT(Rmax, N)
    dht = DiscreteHankelTransform(Rmax, N)
    ctx = dht.context()
    s0 = float(np.log(1e5))
    # seeds on the oracle
    mu, _, _, _ = O.gaussian_model(D, M, j, np.ones(N))
    pI = np.max(D.transform(mu) ** 2) * (D.q / D.q[0]) ** -2
    mu, _, _, _ = O.gaussian_model(D, M, j, pI)
    s = np.log(np.maximum(mu, 1e-3 * mu.max())) - s0
    p_seed = np.max(D.transform(s) ** 2) * (D.q / D.q[0]) ** -4
    t = time.time()
    o = O.lognormal_map(D, M, j, p_seed, s, s0)
    to = time.time() - t
    s_map, Dinv = np.empty(N), np.empty((N, N))
    stats = (ctypes.c_int64 * 9)()
    t = time.time()
    _lib.check(_lib.lib.fh_lognormal_model(ctx, _lib.ptr(_lib.f8(M)), _lib.ptr(_lib.f8(j)), _lib.ptr(_lib.f8(p_seed)),
                                           _lib.ptr(_lib.f8(s)), s0, _lib.ptr(s_map), _lib.ptr(Dinv), stats))
    tg = time.time() - t
    print("N=%d MAP solve: oracle %.3fs %s | gpu %.3fs %s" % (N, to, o["stats"], tg, list(stats)))
    print("   max|ds| = %.3e   Dinv rel = %.3e" % (np.max(np.abs(s_map - o["s"])),
                                                  np.max(np.abs(Dinv - o["Dinv"])) / np.max(np.abs(o["Dinv"]))))
    if not full:
        return
    t = time.time()
    of = O.frank_fit_lognormal(N, Rmax, M, j, alpha=alpha, wsmooth=ws, max_iter=max_iter, diagnostics=True)
    to = time.time() - t
    sg, pg = np.empty(N), np.empty(N)
    dp, ds = np.zeros((max_iter + 1, N)), np.zeros((max_iter + 1, N))
    niter = ctypes.c_int(0)
    t = time.time()
    _lib.check(_lib.lib.fh_fit_lognormal(ctx, _lib.ptr(_lib.f8(M)), _lib.ptr(_lib.f8(j)), alpha, 1e-35, ws, 1e-3,
                                         max_iter, 1e5, _lib.ptr(sg), _lib.ptr(pg), ctypes.byref(niter), None, stats,
                                         _lib.ptr(dp), _lib.ptr(ds)))
    tg = time.time() - t
    Ig, Io = np.exp(sg + s0), of["I"]
    print("   fit alpha=%g ws=%g: oracle %.2fs niter %d totals %s hist %s" % (alpha, ws, to, of["niter"], of["totals"],
                                                                             of["status_hist"]))
    print("                        gpu    %.2fs niter %d stats %s" % (tg, niter.value, list(stats)))
    print("   final I: elementwise %.3e  rel-to-max %.3e   p rel %.3e" % (
        np.max(np.abs(Ig / Io - 1)), np.max(np.abs(Ig - Io)) / Io.max(), np.max(np.abs(pg / of["p"] - 1))))
    for k in (0, 1, 2, 5, 10, 50):
        if k < min(niter.value, of["niter"]):
            print("   iter %3d: p rel %.3e  s abs %.3e" % (k, np.max(np.abs(dp[k] / of["diag_p"][k] - 1)),
                                                          np.max(np.abs(ds[k] - of["diag_s"][k]))))


if __name__ == "__main__":
    cases = [(40, 5000, 1.3, 1e-2), (80, 20000, 1.05, 1e-4)]
    if len(sys.argv) > 1:
        cases = [(int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4]))]
    mi = int(sys.argv[5]) if len(sys.argv) > 5 else 2000
    for c in cases:
        run(*c, max_iter=mi)

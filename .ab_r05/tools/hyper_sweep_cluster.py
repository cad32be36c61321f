"""The fit loop on a cluster of workgroups against the one-workgroup kernel over hyper-parameter extremes (alpha, p0, w_smooth,
tol, max_iter) at N = 130, 300 and 400 -- including the combinations whose Cholesky breaks down or whose power spectrum goes bad:
the same return code, the same iteration count, the same bits.   python3 tools/hyper_sweep_cluster.py"""
import ctypes
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import FixedGeometry, FrankFitter, _lib  # noqa: E402
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities  # noqa: E402

u, v, V, w = mock_disc_visibilities(60000, seed=31, noise_seed=32)
bad, n, rcs = [], 0, {}
for N in (130, 300, 400):
    FF = FrankFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), verbose=False, check_qbounds=False)
    pre = FF.preprocess_visibilities(u, v, V, w)
    M, j = np.ascontiguousarray(pre["M"]), np.ascontiguousarray(pre["j"])
    ctx = FF._DHT.context()
    combos = list(itertools.product((1.0001, 1.05, 2.0, 10.0), (0.0, 1e-30, 1e-15, 1e-5), (0.0, 1e-6, 1e-2, 1.0, 1e3)))
    extra = [(1.05, 1e-15, 1e-4, 1e-3, 0), (1.05, 1e-15, 1e-4, 1e-3, 1), (1.05, 1e-15, 1e-4, 1e-8, 150), (1.05, 1e-15, 1e-4, 0.5, 2000)]
    for c in [(a, p0, ws, 1e-3, 120) for (a, p0, ws) in combos] + extra:
        a, p0, ws, tol, mi = c
        out = []
        for cl in ("1", None):
            if cl is None:
                os.environ.pop("FRANK_AMD_K2_CLUSTER", None)
            else:
                os.environ["FRANK_AMD_K2_CLUSTER"] = cl
            mu, p, nit = np.full(N, np.nan), np.full(N, np.nan), ctypes.c_int()
            rc = _lib.lib.fh_fit_normal(ctx, _lib.ptr(M), _lib.ptr(j), a, p0, ws, tol, mi, _lib.ptr(mu), _lib.ptr(p), ctypes.byref(nit), None, None)
            out.append((rc, nit.value, mu, p))
        (r1, n1, m1, p1), (r2, n2, m2, p2) = out
        ok = r1 == r2 and n1 == n2 and np.array_equal(m1, m2, equal_nan=True) and np.array_equal(p1, p2, equal_nan=True)
        rcs[r1] = rcs.get(r1, 0) + 1
        n += 1
        if not ok:
            bad.append((N,) + c)
            print("N=%d alpha=%g p0=%g ws=%g tol=%g max_iter=%d: one workgroup rc %d niter %d | cluster rc %d niter %d  <-- MISMATCH" % ((N,) + c + (r1, n1, r2, n2)), flush=True)
wg, fb = ctypes.c_int(0), ctypes.c_int64(0)
print("combinations checked: %d (return codes %s), mismatches: %d" % (n, rcs, len(bad)))

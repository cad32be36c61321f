"""The fused fit loop against the library loop over MANY basis sizes (a size-dependent failure hid for two rounds at
N = 127 mod 128): same M, j; fh_fit_normal must return FH_OK on both, the same number of iterations, profiles to 1e-7.
    python3 tools/size_sweep.py [first last step]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import FixedGeometry, FrankFitter, _lib  # noqa: E402
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities  # noqa: E402

a = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else [5, 420, 7]
sizes = sorted(set(list(range(a[0], a[1], a[2])) + [k + d for k in (16, 32, 64, 128, 192, 256, 304, 320, 336, 384, 448, 480, 512, 576, 640, 768, 896, 1008, 1022) for d in (-1, 0, 1) if a[0] <= k + d <= a[1]]))
sizes = [n for n in sizes if 3 <= n <= 1023]
u, v, V, w = mock_disc_visibilities(60000, seed=31, noise_seed=32)
kw = dict(alpha=1.3, weights_smooth=1e-2, verbose=False, check_qbounds=False)
bad = []
for N in sizes:
    FF = FrankFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), **kw)
    pre = FF.preprocess_visibilities(u, v, V, w)
    M, j = np.ascontiguousarray(pre["M"]), np.ascontiguousarray(pre["j"])

    def run(F):
        mu, p, nit = np.empty(N), np.empty(N), ctypes.c_int()
        rc = _lib.lib.fh_fit_normal(F._DHT.context(), _lib.ptr(M), _lib.ptr(j), 1.3, 1e-15, 1e-2, 1e-3, 2000, _lib.ptr(mu),
                                    _lib.ptr(p), ctypes.byref(nit), None, None)
        return rc, nit.value, mu

    rc, nit, mu = run(FF)
    os.environ["FRANK_AMD_K2"] = "rocsolver"
    rc_l, nit_l, mu_l = run(FrankFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), **kw))
    del os.environ["FRANK_AMD_K2"]
    err = np.abs(mu - mu_l).max() / np.abs(mu_l).max() if rc == 0 and rc_l == 0 else float("nan")
    ok = rc == 0 and rc_l == 0 and nit == nit_l and err < 1e-7
    if not ok:
        bad.append(N)
    print("N=%3d  fused rc %d niter %d | library rc %d niter %d | %.1e %s" % (N, rc, nit, rc_l, nit_l, err, "" if ok else "  <-- MISMATCH"), flush=True)
print("sizes checked: %d, mismatches: %s" % (len(sizes), bad))

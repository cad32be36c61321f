"""The deferred trailing update of the fit loop (fit_loop.hip, solve_posterior<0, 4>, FRANK_AMD_K2_RR) against the kernel of
memory over the basis sizes it covers: mu, p and the iteration count must be the same bits.
    python3 tools/rr_sweep.py [first=47] [last=303] [step=3]
"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["FRANK_AMD_K2_CLUSTER"] = "1"
from frank_amd import FixedGeometry, FrankFitter, _lib  # noqa: E402
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 47
last = int(sys.argv[2]) if len(sys.argv) > 2 else 303
step = int(sys.argv[3]) if len(sys.argv) > 3 else 3
u, v, V, w = mock_disc_visibilities(100000, seed=31, noise_seed=32)
bad = 0
sizes = list(range(first, last + 1, step))
for N in sizes:
    FF = FrankFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), verbose=False)
    m = FF.preprocess_visibilities(u, v, V, w)
    M, j = np.ascontiguousarray(m["M"]), np.ascontiguousarray(m["j"])
    ctx = FF._DHT.context()
    res = []
    for d in ("0", "1"):  # (the other forms first)
        os.environ["FRANK_AMD_K2_RR"] = d
        mu, p, nit = np.empty(N), np.empty(N), ctypes.c_int(0)
        rc = _lib.lib.fh_fit_normal(ctx, _lib.ptr(M), _lib.ptr(j), 1.05, 1e-15, 1e-4, 1e-3, 80, _lib.ptr(mu), _lib.ptr(p),
                                    ctypes.byref(nit), None, None)
        res.append((rc, nit.value, mu, p))
    same = res[0][0] == res[1][0] and res[0][1] == res[1][1] and np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])
    if not same:
        bad += 1
        print("N = %d DIFFERS: rc %d / %d, passes %d / %d, max|dmu| %.2e" % (N, res[0][0], res[1][0], res[0][1], res[1][1],
                                                                          float(np.abs(res[0][2] - res[1][2]).max())), flush=True)
print("%d sizes %d .. %d: %d differ" % (len(sizes), sizes[0], sizes[-1], bad))
sys.exit(1 if bad else 0)

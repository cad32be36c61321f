"""BASELINE configs[2] (N=300, 1e7 visibilities, LogNormal) on the resident table: seconds, Newton counters and, with
FRANK_AMD_LIB pointing at the `make timing` build, the in-kernel phase timers (printed to stderr by the library).
    python tools/ln_fullsize.py [nvis [linear|reference]]
"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from frank_amd import _lib as L  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

nvis = int(float(sys.argv[1])) if len(sys.argv) > 1 else bench.N_VIS
f = bench.Fitter(L, bench.N_COLL, 0)
f.nfit = nvis
f.upload(*mock_disc_visibilities(nvis, seed=0, noise_seed=50))
N = bench.N_COLL
h = bench.HYPER
H0, qmn, qmx = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
L.set_lognormal_linesearch(f.ctx, sys.argv[2] if len(sys.argv) > 2 else "linear")
for rep in range(2):
    s_map, p = np.empty(N), np.empty(N)
    nit = ctypes.c_int(0)
    stats = (ctypes.c_int64 * 9)()
    f.bin()
    L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 1, None, None, ctypes.byref(H0), ctypes.byref(qmn),
                                    ctypes.byref(qmx)))
    t1 = time.perf_counter()
    L.check(L.lib.fh_fit_lognormal(f.ctx, None, None, 1.3, 1e-35, 1e-2, h["tol"], h["max_iter"], 1e5, L.ptr(s_map),
                                   L.ptr(p), ctypes.byref(nit), None, stats, None, None))
    dt = time.perf_counter() - t1
    I = np.exp(s_map + np.log(1e5))
    print("fit %.3f s  iterations %d  newton steps %d  evaluations %d  hessians %d  -> %.3f ms/hessian all-in, "
          "%.1f evaluations/step; I in [%.4g, %.4g]" % (dt, nit.value, stats[1], stats[2], stats[3],
                                                       1e3 * dt / max(stats[3], 1), stats[2] / max(stats[1], 1),
                                                       I.min(), I.max()), flush=True)

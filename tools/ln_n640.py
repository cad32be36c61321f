"""bench.py's extra.lognormal_N640 leg alone: N = 640 on the 1e7-visibility mock table, LogNormal, alpha = 1.3, w_smooth = 1e-2, at most
200 passes -- through the persistent kernel's WIDE form (default) or the host-driven route (FRANK_AMD_LN_WIDE=host).
   python3 tools/ln_n640.py [N]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from frank_amd import _lib as L  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 640
f2 = bench.Fitter(L, N, 0)
f2.nfit = 10_000_000
f2.upload(*mock_disc_visibilities(f2.nfit, seed=0, noise_seed=50))
for rep in range(2):
    s2, p2 = np.empty(N), np.empty(N)
    nit2 = ctypes.c_int(0)
    st2 = (ctypes.c_int64 * 9)()
    H0, qmn, qmx = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    f2.bin()
    L.check(L.lib.fh_stats_finalize(f2.ctx, ctypes.byref(f2.geom), 0, 1, None, None, ctypes.byref(H0), ctypes.byref(qmn), ctypes.byref(qmx)))
    t1 = time.perf_counter()
    L.check(L.lib.fh_fit_lognormal(f2.ctx, None, None, 1.3, 1e-35, 1e-2, 1e-3, 200, 1e5, L.ptr(s2), L.ptr(p2), ctypes.byref(nit2), None, st2, None, None))
    t2 = time.perf_counter()
    I2 = np.exp(s2 + np.log(1e5))
    print("N = %d (%s): fit %.3f s, %d passes = %.2f ms per pass; Newton steps %d, evaluations %d, Hessians %d; I in [%.4g, %.4g]" % (
        N, os.environ.get("FRANK_AMD_LN_WIDE", "kernel"), t2 - t1, nit2.value, 1e3 * (t2 - t1) / max(nit2.value, 1), st2[1], st2[2], st2[3], I2.min(), I2.max()), flush=True)

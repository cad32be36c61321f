"""The driver's K-step region taken apart: host time of K submissions (binning + finalisation + hand-over, no fit launched),
the moment the binning stream has finished them, and the end of the region (the K fits collected).
   python tools/submit_phase.py [K]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from frank_amd import _lib as L  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
f = bench.Fitter(L, 300, 0)
u, v, V, w = mock_disc_visibilities(10_000_000, seed=1, noise_seed=2)
f.upload(u, v, V, w)
for _ in range(3):
    f.fit()
f.run_steps(5)
for rep in range(3):
    L.check(L.lib.fh_ctx_synchronize(f.ctx)) if hasattr(L.lib, "fh_ctx_synchronize") else None
    t0 = time.perf_counter()
    tickets = [f.submit() for _ in range(K)]
    t1 = time.perf_counter()
    nit = [f.collect(t) for t in tickets]
    t2 = time.perf_counter()
    print("K = %d: submissions returned after %.2f ms (%.3f ms each), region %.2f ms -> %.1f fits/s; the fits alone %.2f ms (%d passes)" % (
        K, 1e3 * (t1 - t0), 1e3 * (t1 - t0) / K, 1e3 * (t2 - t0), K / (t2 - t0), 1e3 * (t2 - t1), nit[-1]))

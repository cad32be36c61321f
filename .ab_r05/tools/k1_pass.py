"""Binning pass of 1e7 visibilities at N = 300 (default path): time per pass from back-to-back passes and from the events of
the last one; M, j of the pass against the fixture of the reference when it is there.   python3 tools/k1_pass.py [n] [N] [reps]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from frank_amd import _lib as L  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
f = bench.Fitter(L, N, 0)
u, v, V, w = mock_disc_visibilities(n, seed=0, noise_seed=50)
f.upload(u, v, V, w)
for _ in range(3):
    f.bin()
f.sync()
t0 = time.perf_counter()
for _ in range(reps):
    f.bin()
f.sync()
dt = (time.perf_counter() - t0) / reps
print("tag=%s n=%d N=%d: %.4f ms per pass (back to back), events: pre-pass %.4f ms + Gram %.4f ms" % (
    os.environ.get("K1_TAG", "default"), n, N, dt * 1e3, f.prepass_ms(), f.kernel_ms()))
print("   = %.0f GB/s of the 40 B per visibility" % (40 * n / dt / 1e9))
M, j = np.empty((N, N)), np.empty(N)
H0, qmn, qmx = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 1, L.ptr(M), L.ptr(j), ctypes.byref(H0), ctypes.byref(qmn),
                                ctypes.byref(qmx)))
fx = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "fit_N300_1e7.npz")
if n == 10_000_000 and N == 300 and os.path.exists(fx):
    g = np.load(fx)
    print("   vs the reference's fixture: M %.2e  j %.2e  H0 %.2e (relative to the maximum)" % (
        np.abs(M - g["M"]).max() / np.abs(g["M"]).max(), np.abs(j - g["j"]).max() / np.abs(g["j"]).max(),
        abs(H0.value - float(g["H0"])) / abs(float(g["H0"]))))
print("   qmin %.6e qmax %.6e  sum|M| %.17e" % (qmn.value, qmx.value, np.abs(M).sum()))

"""The fused form of the moments pre-pass (bin_fused.hip, FRANK_AMD_K1_FUSED=1) against the sorted one (bin_prepass.hip), one
process, the same resident table: ms per pass (back to back; histograms kept / looked at again), M, j, H0 of one against the
other and against the reference's fixture, and whether two passes of each form land on the same bits.
    python3 tools/k1_fused.py [n] [N] [reps] [stretch]"""
import ctypes
import hashlib
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
stretch = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
f = bench.Fitter(L, N, 0)
u, v, V, w = mock_disc_visibilities(n, seed=0, noise_seed=50)
if stretch != 1.0:
    u, v = u * stretch, v * stretch
f.upload(u, v, V, w)


def stats():
    M, j = np.empty((N, N)), np.empty(N)
    H0, qmn, qmx = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 1, L.ptr(M), L.ptr(j), ctypes.byref(H0), ctypes.byref(qmn),
                                    ctypes.byref(qmx)))
    return M, j, H0.value


def measure(tag, env):
    for k in ("FRANK_AMD_K1_FUSED", "FRANK_AMD_K1_NO_HIST_CACHE", "FRANK_AMD_NO_RANGE_CACHE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    L.check(L.lib.fh_ctx_reload_env(f.ctx))
    for _ in range(3):
        f.bin()
    f.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        f.bin()
    f.sync()
    dt = (time.perf_counter() - t0) / reps
    pre, gram = f.prepass_ms(), f.kernel_ms()
    M, j, H0 = stats()
    f.bin()
    M2, j2, H02 = stats()
    same = bool(np.array_equal(M, M2) and np.array_equal(j, j2) and H0 == H02)
    print("%-44s %.4f ms per pass (events: pre-pass %.4f + Gram %.4f) = %4.0f GB/s of 40 B/row; two passes same bits: %s  sha %s" % (
        tag, dt * 1e3, pre, gram, 40 * n / dt / 1e9, same, hashlib.sha1(M.tobytes() + j.tobytes()).hexdigest()[:10]))
    return M, j, H0, dt


print("n=%d N=%d stretch=%g" % (n, N, stretch))
Ms, js, H0s, ts = measure("sorted, histograms kept", {})
measure("sorted, every pass looks at (u, v) again", {"FRANK_AMD_K1_NO_HIST_CACHE": "1"})
Mf, jf, H0f, tf = measure("FUSED, histograms kept", {"FRANK_AMD_K1_FUSED": "1"})
measure("FUSED, every pass looks at (u, v) again", {"FRANK_AMD_K1_FUSED": "1", "FRANK_AMD_K1_NO_HIST_CACHE": "1"})
print("fused against sorted: M %.2e  j %.2e  H0 %.2e (relative to the maximum)" % (
    np.abs(Mf - Ms).max() / np.abs(Ms).max(), np.abs(jf - js).max() / np.abs(js).max(), abs(H0f - H0s) / abs(H0s)))
fx = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "fit_N300_1e7.npz")
if n == 10_000_000 and N == 300 and stretch == 1.0 and os.path.exists(fx):
    g = np.load(fx)
    for tag, (M, j, H0) in (("sorted", (Ms, js, H0s)), ("fused", (Mf, jf, H0f))):
        print("%s vs the reference's fixture: M %.2e  j %.2e  H0 %.2e" % (
            tag, np.abs(M - g["M"]).max() / np.abs(g["M"]).max(), np.abs(j - g["j"]).max() / np.abs(g["j"]).max(),
            abs(H0 - float(g["H0"])) / abs(float(g["H0"]))))

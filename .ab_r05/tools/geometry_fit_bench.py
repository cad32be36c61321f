"""The geometry fits at size: seconds per fit, residual evaluations, and what one evaluation is made of.
    python3 tools/geometry_fit_bench.py [nvis [device|scipy]]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import DiscreteHankelTransform, _lib  # noqa: E402
from frank_amd.constants import rad_to_arcsec  # noqa: E402
from frank_amd.geometry import FitGeometryFourierBessel, FitGeometryGaussian, _ResidentTable  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 6
OPTS = (sys.argv[2],) if len(sys.argv) > 2 else ("device", "scipy")
u, v, V, w = mock_disc_visibilities(n, seed=71, noise_seed=72, weight=1e6, qmax=1e6)


class Counting(FitGeometryFourierBessel):
    calls = 0


    def _trial_geometry(self, params):  # (called once per residual evaluation under either optimiser)
        Counting.calls += 1
        return FitGeometryFourierBessel._trial_geometry(self, params)


for opt in OPTS:
    for rep in range(2):
        Counting.calls = 0
        t0 = time.perf_counter()
        f = Counting(2.0, 20, guess=[30.0, 80.0, 0.0, 0.0], optimizer=opt)
        f.fit(u, v, V, w)
        dt = time.perf_counter() - t0
        print("FitGeometryFourierBessel optimizer=%s n=%d: %.3f s, %d residual evaluations (%.1f ms each all-in) -> inc %.5f PA %.5f dRA %.7f dDec %.7f"
              % (opt, n, dt, Counting.calls, 1e3 * dt / Counting.calls, f.inc, f.PA, f.dRA, f.dDec), flush=True)
    for rep in range(2):
        t0 = time.perf_counter()
        g = FitGeometryGaussian(guess=[30.0, 80.0, 0.0, 0.0], optimizer=opt)
        g.fit(u, v, V, w)
        print("FitGeometryGaussian      optimizer=%s n=%d: %.3f s -> inc %.5f PA %.5f dRA %.7f dDec %.7f" % (opt, n, time.perf_counter() - t0, g.inc, g.PA, g.dRA, g.dDec), flush=True)

# one evaluation, piece by piece
DHT = DiscreteHankelTransform(2.0 / rad_to_arcsec, 20)
t = _ResidentTable(DHT.device, u, v, V, w)
ctx, N = DHT.context(), 20
gg = _lib.fh_geometry(30.0, 80.0, 0.0, 0.0)
M, j, I = np.empty((N, N)), np.empty(N), np.empty(N)
H0, a, b, sv, ss = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int(), ctypes.c_double()
out = np.empty(2 * n)


def timed(fn, reps=10):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return 1e3 * (time.perf_counter() - t0) / reps


def bin_():
    _lib.check(_lib.lib.fh_bin_reset(ctx))
    _lib.check(_lib.lib.fh_bin_visibilities(ctx, ctypes.byref(gg), t.handle, 0, n))
    _lib.check(_lib.lib.fh_stats_finalize(ctx, ctypes.byref(gg), 0, 0, _lib.ptr(M), _lib.ptr(j), ctypes.byref(H0), ctypes.byref(a), ctypes.byref(b)))


def solve_():
    _lib.check(_lib.lib.fh_gaussian_model(ctx, _lib.ptr(M), _lib.ptr(j), None, _lib.ptr(I), None, None, ctypes.byref(sv)))


def resid_(o):
    _lib.check(_lib.lib.fh_vis_residuals(ctx, ctypes.byref(gg), 0, t.handle, 0, n, _lib.ptr(I), _lib.ptr(o) if o is not None else None, ctypes.byref(ss)))


tb, ts = timed(bin_), timed(solve_)
tk, tr = timed(lambda: resid_(None)), timed(lambda: resid_(out), 3)
print("one evaluation at n=%d: bin + finalize %.3f ms, solve %.3f ms, residual kernel (sum of squares only) %.3f ms = %.0f GB/s of 40 B/row, "
      "with the 16 B/row written and copied to the host %.2f ms" % (n, tb, ts, tk, 40e-9 * n / (tk * 1e-3), tr))

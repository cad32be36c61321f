"""The geometry fits with the device-side Levenberg-Marquardt against the same fits under SciPy's driver (the reference's), over
table sizes, basis sizes, true geometries, what is pinned, scalar / uneven weights.  Both must end in a minimum of the same
depth: the chi^2 of the Fourier-Bessel fit at the two fitted geometries within 1e-7 (relative); for the Gaussian the fitted
inc within 1e-3 deg, PA within 1e-3 deg / sin(inc) (a face-on disc has no position angle), the phase centre within 1e-6 arcsec.
    python3 tools/geometry_fit_sweep.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import DiscreteHankelTransform, FixedGeometry, _lib  # noqa: E402
from frank_amd.constants import rad_to_arcsec  # noqa: E402
from frank_amd.geometry import FitGeometryFourierBessel, FitGeometryGaussian, _ResidentTable  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

bad, done = [], 0
for truth in ((34.97, 85.76, 1.9e-3, 2.5e-3), (10.0, 120.0, -0.02, 0.01), (70.0, 5.0, 0.05, -0.03), (50.0, 179.0, 0.0, 0.0)):
    geom = dict(inc=truth[0], PA=truth[1], dRA=truth[2], dDec=truth[3])
    for n in (2000, 65537, 300001):
        u, v, V, w = mock_disc_visibilities(n, seed=n % 97, noise_seed=7, weight=1e6, qmax=2.2e6, geometry=geom)
        for wform in ("array", "scalar", "uneven"):
            ww = w if wform == "array" else (float(w[0]) if wform == "scalar" else w * np.random.default_rng(1).uniform(0.3, 3.0, n))
            if wform != "array" and n != 65537:
                continue
            for pins in ({}, dict(inc_pa=truth[:2]), dict(phase_centre=truth[2:])):
                guess = [truth[0] + 4.0, truth[1] - 5.0, 0.0, 0.0]
                fits = [("gauss", lambda opt: FitGeometryGaussian(guess=list(guess), optimizer=opt, **pins))]
                for N in ((20,) if n != 65537 or wform != "array" else (10, 20, 40)):
                    fits.append(("fb N=%d" % N, lambda opt, N=N: FitGeometryFourierBessel(2.0, N, guess=list(guess), optimizer=opt, **pins)))
                for name, make in fits:
                    res = []
                    for opt in ("device", "scipy"):
                        f = make(opt)
                        try:
                            f.fit(u, v, V, ww)
                            res.append(np.array([f.inc, f.PA, f.dRA, f.dDec]))
                        except RuntimeError as e:
                            res.append(None)
                    done += 1
                    if res[0] is None or res[1] is None:
                        ok = res[0] is None and res[1] is None
                        d = "fail/fail" if ok else "ONE FAILED"
                    else:
                        dPA = abs(res[0][1] - res[1][1])
                        dPA = min(dPA, 180 - dPA) * np.sin(np.deg2rad(max(res[0][0], res[1][0])))
                        if name.startswith("fb"):
                            N = int(name.split("=")[1])
                            DHT = DiscreteHankelTransform(2.0 / rad_to_arcsec, N)
                            t = _ResidentTable(DHT.device, u, v, V, np.broadcast_to(ww, u.shape))
                            ss = []
                            for x in res:
                                gg, I = FitGeometryFourierBessel._profile_under(FixedGeometry(*x), DHT, t)
                                c = ctypes.c_double()
                                _lib.check(_lib.lib.fh_vis_residuals(DHT.context(), ctypes.byref(gg), 0, t.handle, 0, n, _lib.ptr(I), None, ctypes.byref(c)))
                                ss.append(c.value)
                            t.close()
                            ok = abs(ss[0] / ss[1] - 1) < 1e-7
                            d = "chi2 ratio - 1 = %+.1e  d(inc) %.1e d(PA) sin(inc) %.1e d(phase) %.1e -> %s" % (
                                ss[0] / ss[1] - 1, abs(res[0][0] - res[1][0]), dPA, np.abs(res[0][2:] - res[1][2:]).max(), np.array2string(res[0], precision=4))
                        else:
                            ok = abs(res[0][0] - res[1][0]) < 1e-3 and dPA < 1e-3 and np.abs(res[0][2:] - res[1][2:]).max() < 1e-6
                            d = "d(inc) %.1e d(PA) sin(inc) %.1e d(phase) %.1e -> %s" % (abs(res[0][0] - res[1][0]), dPA, np.abs(res[0][2:] - res[1][2:]).max(),
                                                                                       np.array2string(res[0], precision=4))
                    if not ok:
                        bad.append((truth, n, wform, tuple(pins), name))
                    print("truth %s n=%6d w=%-6s pins=%-14s %-8s %s%s" % (truth[:2], n, wform, ",".join(pins) or "-", name, d, "" if ok else "  <-- MISMATCH"), flush=True)
print("fits compared: %d, mismatches: %s" % (done, bad))
sys.exit(1 if bad else 0)

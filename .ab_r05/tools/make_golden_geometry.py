#!/usr/bin/env python3
"""Geometry-fit fixture (run in the BUILD container only; imports the reference).

    python3 tools/make_golden_geometry.py

Mock-disc visibilities (frank_amd.mock, 20 000 rows, the mock geometry inc=34.97, PA=85.76, dRA=1.9e-3, dDec=2.5e-3) through
the reference's two geometry fits (geometry.py:404-763), started at (30, 80, 0, 0), in their three call forms (everything free, inc/PA given, phase centre
given), and its residual function FitGeometryFourierBessel._residual at one trial geometry (every 8th entry + the sum of
squares).  Stored: inputs' recipe (seeds), the fitted (inc, PA, dRA, dDec), the residual samples.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, ".."))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, ROOT)

import scipy  # noqa: E402
import frank  # noqa: E402
from frank.geometry import FitGeometryFourierBessel, FitGeometryGaussian  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402  (table-driven generator: NumPy only, no library)

N_VIS, SEED, NOISE_SEED = 20000, 71, 72
RMAX, N = 2.0, 20
TRIAL = (30.0, 80.0, 0.01, -0.005)
WEIGHT, QMAX = 1e6, 1e6  # (noise 1 mJy per visibility: the geometry is well constrained)
GUESS = [30.0, 80.0, 0.0, 0.0]


def main():
    u, v, V, w = mock_disc_visibilities(N_VIS, seed=SEED, noise_seed=NOISE_SEED, weight=WEIGHT, qmax=QMAX)
    w = np.full(u.size, w) if np.ndim(w) == 0 else w
    sha = hashlib.sha256(b"".join(np.ascontiguousarray(a).tobytes() for a in (u, v, V, w))).hexdigest()
    out = dict(weight=WEIGHT, qmax=QMAX, guess=np.array(GUESS), n=N_VIS, seed=SEED, noise_seed=NOISE_SEED, Rmax=RMAX, N=N, input_sha256=sha, trial=np.array(TRIAL))
    cases = {"free": {}, "incpa": dict(inc_pa=(34.97, 85.76)), "phase": dict(phase_centre=(1.9e-3, 2.5e-3))}
    for tag, kw in cases.items():
        g = FitGeometryGaussian(guess=list(GUESS), **kw)
        g.fit(u, v, V, w)
        out["gauss_" + tag] = np.array([g.inc, g.PA, g.dRA, g.dDec])
        print("gauss", tag, out["gauss_" + tag])
        f = FitGeometryFourierBessel(RMAX, N, guess=list(GUESS), **kw)
        f.fit(u, v, V, w)
        out["fb_" + tag] = np.array([f.inc, f.PA, f.dRA, f.dDec])
        print("fourier-bessel", tag, out["fb_" + tag])
    g = FitGeometryGaussian()  # the default starting point (10, 10, 0, 0)
    g.fit(u, v, V, w)
    out["gauss_default_guess"] = np.array([g.inc, g.PA, g.dRA, g.dDec])
    print("gauss default guess", out["gauss_default_guess"])
    f = FitGeometryFourierBessel(RMAX, N)
    r = f._residual(TRIAL, uvdata=[u, v, V, w ** 0.5])
    out["resid_every8"] = r[::8]
    out["resid_sumsq"] = float(np.sum(r * r))
    out.update(meta_reference_version=frank.__version__, meta_numpy=np.__version__, meta_scipy=scipy.__version__)
    path = os.path.join(ROOT, "tests", "golden", "geometry_fits_2e4.npz")
    np.savez_compressed(path, **out)
    print("wrote %s %.1f KB" % (path, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()

"""Seeded synthetic mock-disc visibilities.

The recipe is the reference's own mock-data tutorial (docs/tutorials/mock_data.ipynb
cells 8, 16-17; frank/utilities.py:962-1038): a four-Gaussian ringed profile, sampled
at deprojected baselines through an N=500 DHT, scaled by cos(inc), re-phased by
(dRA, dDec), plus N(0, w^-1/2) noise on the real and imaginary parts.

The noiseless curve V(q) is read from mock_disc_vis_table.npz beside this module (tabulated
once from the reference by tools/make_mock_table.py) and linearly interpolated, so the
generator is NumPy-only and bit-reproducible wherever numpy's default_rng is.
"""
import os

import numpy as np

from frank_amd.constants import rad_to_arcsec, deg_to_rad

# frank/tests.py:141-142 (the AS 209 geometry used throughout the reference tests)
MOCK_GEOMETRY = dict(inc=34.97, PA=85.76, dRA=1.9e-3, dDec=2.5e-3)

_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mock_disc_vis_table.npz")
_cache = {}


def _table(path=None):
    path = path or _TABLE
    if path not in _cache:
        t = np.load(path)
        _cache[path] = (t["q"], t["V"])
    return _cache[path]


def mock_disc_visibilities(n, seed=0, noise_seed=50, weight=400.0, qmin=1e4, qmax=2e6, geometry=None,
                           table=None):
    """Return (u, v, V, w): n synthetic visibilities of the mock disc.

    u, v : float64 [lambda], log-uniform baseline length in [qmin, qmax], uniform angle (seed)
    V    : complex128 [Jy], noisy, NOT phase-centred (the fit removes dRA, dDec)
    w    : float64 [Jy^-2], constant `weight`
    """
    g = dict(MOCK_GEOMETRY if geometry is None else geometry)
    rng = np.random.default_rng(seed)
    q = np.exp(rng.uniform(np.log(qmin), np.log(qmax), n))
    th = rng.uniform(0.0, 2 * np.pi, n)
    u = q * np.cos(th)
    v = q * np.sin(th)
    # deprojected baseline (geometry.py:111-131)
    ct, st = np.cos(g["PA"] * deg_to_rad), np.sin(g["PA"] * deg_to_rad)
    up = (u * ct - v * st) * np.cos(g["inc"] * deg_to_rad)
    vp = u * st + v * ct
    qd = np.hypot(up, vp)
    tq, tV = _table(table)
    Vm = np.interp(qd, tq, tV)
    # forward phase shift (geometry.py:69-77): the source sits (dRA, dDec) off the phase centre
    phi = (u * g["dRA"] + v * g["dDec"]) * (2.0 * np.pi / rad_to_arcsec)
    V = Vm * (np.cos(phi) + 1j * np.sin(phi))
    nrng = np.random.default_rng(noise_seed)
    sig = weight ** -0.5
    V = V + sig * nrng.standard_normal(n) + 1j * sig * nrng.standard_normal(n)
    w = np.full(n, float(weight))
    return u, v, V, w

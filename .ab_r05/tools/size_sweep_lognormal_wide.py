"""LogNormalMAPModel beyond the persistent kernel (lognormal_wide.hip) against the CPU oracle over basis sizes 321 ... 1023: the
profile, the Hessian count (reference line search) and the step count.   python tools/size_sweep_lognormal_wide.py"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from frank_amd import DiscreteHankelTransform, LogNormalMAPModel  # noqa: E402
from frank_amd.constants import rad_to_arcsec  # noqa: E402
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities  # noqa: E402
from oracle import oracle as fo  # noqa: E402  (the referee of this development sweep)

g = MOCK_GEOMETRY
GEOM = (g["inc"], g["PA"], g["dRA"], g["dDec"])
bad = 0
for N, rmax_as, nvis in ((321, 2.0, 400000), (352, 1.8, 300000), (448, 1.0, 100000), (511, 1.0, 150000), (512, 0.9, 150000),
                         (640, 0.8, 150000), (800, 0.6, 150000), (1023, 0.5, 150000)):
    rmax = rmax_as / rad_to_arcsec
    u, v, V, w = mock_disc_visibilities(nvis, seed=51, noise_seed=52)
    m = fo.map_visibilities(N, rmax, GEOM, u, v, V, w, check_qbounds=False)
    D = fo.DHT(rmax, N)
    s0 = float(np.log(1e5))
    mu, _, _, _ = fo.gaussian_model(D, m["M"], m["j"], np.ones(N))
    pI = np.max(D.transform(mu) ** 2) * (D.q / D.q[0]) ** -2
    mu, _, _, _ = fo.gaussian_model(D, m["M"], m["j"], pI)
    s_guess = np.log(np.maximum(mu, 1e-3 * mu.max())) - s0
    p_seed = np.max(D.transform(s_guess) ** 2) * (D.q / D.q[0]) ** -4
    ref = fo.lognormal_map(D, m["M"], m["j"], p_seed, s_guess, s0)
    d = DiscreteHankelTransform(rmax, N)
    Iref = np.exp(ref["s"] + s0)
    out = []
    for ls in ("reference", "linear"):
        fit = LogNormalMAPModel(d, m["M"], m["j"], p_seed, guess=s_guess, s0=s0, linesearch=ls)
        out.append((np.abs(np.exp(fit.MAP + s0) - Iref).max() / Iref.max(), fit._newton_stats))
    ok = out[0][0] < 1e-6 and out[1][0] < 1e-6 and (ref["stats"][0] != 0 or out[0][1][3] == ref["stats"][3])
    bad += not ok
    print("N=%4d oracle %s | reference search: profile %.1e steps %d Hessians %d | linear: profile %.1e steps %d Hessians %d%s" % (
        N, tuple(ref["stats"]), out[0][0], out[0][1][1], out[0][1][3], out[1][0], out[1][1][1], out[1][1][3],
        "" if ok else "  <-- MISMATCH"), flush=True)
print("sizes with a difference:", bad)

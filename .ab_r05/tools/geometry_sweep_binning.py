"""The default (moments) binning path against the rows path over source geometries (face-on, edge-on-ish, zero and large phase
centres -- the pre-pass switches to the library's sincos beyond 1e5 rad --, negative angles), fp64 and fp32 tables, scalar
weights: M, j to 1e-12, H0 to 1e-11.   python3 tools/geometry_sweep_binning.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import FixedGeometry, FourierBesselFitter  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

u, v, V, w = mock_disc_visibilities(50000, seed=11, noise_seed=12)
geoms = [dict(inc=0.0, PA=0.0, dRA=0.0, dDec=0.0), dict(inc=30.0, PA=40.0, dRA=0.03, dDec=-0.02), dict(inc=85.0, PA=179.0, dRA=0.0, dDec=0.0),
         dict(inc=-20.0, PA=-135.0, dRA=-0.5, dDec=0.7), dict(inc=60.0, PA=10.0, dRA=30.0, dDec=-45.0), dict(inc=45.0, PA=90.0, dRA=1e-9, dDec=0.0)]
bad = []
for N in (40, 100, 300):
    for gi, g in enumerate(geoms):
        for wmode in ("vector", "scalar"):
            ww = w if wmode == "vector" else np.float64(w.mean())
            res = []
            for mode in ("moments", "rows"):
                if mode == "rows":
                    os.environ["FRANK_AMD_K1"] = "rows"
                F = FourierBesselFitter(2.0, N, FixedGeometry(**g), verbose=False)
                F._vis_map.check_qbounds = False
                res.append(F.preprocess_visibilities(u, v, V, ww))
                os.environ.pop("FRANK_AMD_K1", None)
            a, b = res
            eM = np.abs(a["M"] - b["M"]).max() / np.abs(b["M"]).max()
            ej = np.abs(a["j"] - b["j"]).max() / np.abs(b["j"]).max()
            eH = abs(a["null_likelihood"] - b["null_likelihood"]) / abs(b["null_likelihood"])
            ok = eM < 1e-12 and ej < 1e-12 and eH < 1e-11
            if not ok:
                bad.append((N, gi, wmode))
            print("N=%3d geometry %d weights %-6s  M %.1e  j %.1e  H0 %.1e %s" % (N, gi, wmode, eM, ej, eH, "" if ok else "  <-- MISMATCH"), flush=True)
print("mismatches:", bad)

"""The debris model (vis_model='debris', scale_height) on the fused rows kernel against the rows-to-memory + rocBLAS path
(FRANK_AMD_K1=wide) over basis sizes and scale heights, and single-precision tables against double ones on the default path:
M, j to 1e-12 (debris) / 1e-6 (fp32 table).   python3 tools/debris_sweep.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import FixedGeometry, FourierBesselFitter  # noqa: E402
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities  # noqa: E402

u, v, V, w = mock_disc_visibilities(30000, seed=21, noise_seed=22)
bad = []
for N in (16, 40, 100, 255, 300, 303, 304, 400, 511):
    r = np.linspace(0, 2.0, N)
    for hname, hfun in (("flat 0.02", lambda R: 0.02 * np.ones_like(R)), ("flared", lambda R: 0.05 * (R + 0.1) ** 1.2), ("zero", lambda R: np.zeros_like(R))):
        res = []
        for mode in ("fused", "wide"):
            if mode == "wide":
                os.environ["FRANK_AMD_K1"] = "wide"
            F = FourierBesselFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), verbose=False, assume_optically_thick=False, scale_height=hfun)
            F._vis_map.check_qbounds = False
            res.append(F.preprocess_visibilities(u, v, V, w))
            os.environ.pop("FRANK_AMD_K1", None)
        a, b = res
        eM = np.abs(a["M"] - b["M"]).max() / np.abs(b["M"]).max()
        ej = np.abs(a["j"] - b["j"]).max() / np.abs(b["j"]).max()
        eH = abs(a["null_likelihood"] - b["null_likelihood"]) / abs(b["null_likelihood"])
        ok = eM < 1e-12 and ej < 1e-12 and eH < 1e-12
        if not ok:
            bad.append((N, hname))
        print("debris N=%3d %-9s  M %.1e  j %.1e  H0 %.1e %s" % (N, hname, eM, ej, eH, "" if ok else "  <-- MISMATCH"), flush=True)
print("mismatches:", bad)

"""The default (moments) binning path against the rows path (FRANK_AMD_K1=rows: every visibility through the design-block +
Gram kernel, N <= 511) over many basis sizes: M, j to 1e-12 of their maxima, H0 to 1e-12.
    python3 tools/size_sweep_binning.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import FixedGeometry, FourierBesselFitter  # noqa: E402
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities  # noqa: E402

sizes = list(range(3, 512))
sizes = [n for n in sizes if 3 <= n <= 511]
u, v, V, w = mock_disc_visibilities(40000, seed=3, noise_seed=4)
bad = []
for N in sizes:
    FB = FourierBesselFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), verbose=False, check_qbounds=False) if False else None
    res = []
    for mode in ("moments", "rows"):
        if mode == "rows":
            os.environ["FRANK_AMD_K1"] = "rows"
        F = FourierBesselFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), verbose=False)
        F._vis_map._check_qbounds = False
        res.append(F.preprocess_visibilities(u, v, V, w))
        os.environ.pop("FRANK_AMD_K1", None)
    a, b = res
    eM = np.abs(a["M"] - b["M"]).max() / np.abs(b["M"]).max()
    ej = np.abs(a["j"] - b["j"]).max() / np.abs(b["j"]).max()
    eH = abs(a["null_likelihood"] - b["null_likelihood"]) / abs(b["null_likelihood"])
    ok = eM < 1e-12 and ej < 1e-12 and eH < 1e-12
    if not ok:
        bad.append(N)
    print("N=%3d  M %.1e  j %.1e  H0 %.1e %s" % (N, eM, ej, eH, "" if ok else "  <-- MISMATCH"), flush=True)
print("sizes checked: %d, mismatches: %s" % (len(sizes), bad))

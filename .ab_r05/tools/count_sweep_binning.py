"""The default (moments) binning path against the rows path over many ROW COUNTS (tile and segment boundaries of the pre-pass,
buckets of 0 .. 17 rows): M, j, H0 to 1e-12.   python3 tools/count_sweep_binning.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import FixedGeometry, FourierBesselFitter  # noqa: E402
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities  # noqa: E402

N = 100
U, Vv, V, W = mock_disc_visibilities(300000, seed=9, noise_seed=10)
counts = sorted(set([1, 2, 3, 5, 15, 16, 17, 31, 33, 63, 64, 65, 127, 128, 129, 255, 257, 511, 513, 1000, 1023, 1024, 1025, 2047, 2049, 4095, 4096,
                     4097, 8191, 8192, 8193, 12345, 16383, 16384, 16385, 32767, 32769, 65535, 65536, 65537, 100001, 131071, 131073, 200003,
                     262143, 262145, 300000] + list(range(700, 9000, 613))))
F = {}
for mode in ("moments", "rows"):
    if mode == "rows":
        os.environ["FRANK_AMD_K1"] = "rows"
    F[mode] = FourierBesselFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), verbose=False)
    F[mode]._vis_map.check_qbounds = False
    F[mode].preprocess_visibilities(U[:10], Vv[:10], V[:10], W[:10])  # (the context is created under the mode's environment)
    os.environ.pop("FRANK_AMD_K1", None)
bad = []
for n in counts:
    a = F["moments"].preprocess_visibilities(U[:n], Vv[:n], V[:n], W[:n])
    b = F["rows"].preprocess_visibilities(U[:n], Vv[:n], V[:n], W[:n])
    eM = np.abs(a["M"] - b["M"]).max() / np.abs(b["M"]).max()
    ej = np.abs(a["j"] - b["j"]).max() / np.abs(b["j"]).max()
    eH = abs(a["null_likelihood"] - b["null_likelihood"]) / abs(b["null_likelihood"])
    ok = eM < 1e-12 and ej < 1e-12 and eH < 1e-12
    if not ok:
        bad.append(n)
    print("n=%6d  M %.1e  j %.1e  H0 %.1e %s" % (n, eM, ej, eH, "" if ok else "  <-- MISMATCH"), flush=True)
print("counts checked: %d, mismatches: %s" % (len(counts), bad))

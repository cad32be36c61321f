"""Where the host-array call VisibilityMapping.map_visibilities spends its time at 1e7 rows: upload (allocation + copies), reset,
binning (first sight of the table: range pass + histogram), finalize (M, j to the host), destroy (hipFree), and the whole
Python call beside it.     python3 tools/map_phases.py [n=1e7]
"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import FixedGeometry, FrankFitter, _lib  # noqa: E402
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
u, v, V, w = mock_disc_visibilities(n, seed=0, noise_seed=50)
FF = FrankFitter(2.0, 300, FixedGeometry(**MOCK_GEOMETRY), verbose=False)
FF.preprocess_visibilities(u, v, V, w)
ctx = FF._DHT.context()
g = _lib.make_geometry(FixedGeometry(**MOCK_GEOMETRY))
L = _lib.lib
N = 300
M, j = np.empty((N, N)), np.empty(N)
H0, a, b = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
Vp = V.view(np.float64).ctypes.data_as(ctypes.POINTER(ctypes.c_double))
for rep in range(3):
    t = [time.perf_counter()]
    vis = ctypes.c_void_p()
    _lib.check(L.fh_vis_upload_c128(0, _lib.ptr(u), _lib.ptr(v), Vp, _lib.ptr(w), w.size, n, ctypes.byref(vis)))
    t.append(time.perf_counter())
    _lib.check(L.fh_bin_reset(ctx))
    _lib.check(L.fh_bin_visibilities(ctx, ctypes.byref(g), vis, 0, n))
    t.append(time.perf_counter())
    _lib.check(L.fh_stats_finalize(ctx, ctypes.byref(g), 0, 1, _lib.ptr(M), _lib.ptr(j), ctypes.byref(H0), ctypes.byref(a), ctypes.byref(b)))
    t.append(time.perf_counter())
    L.fh_vis_destroy(vis)
    t.append(time.perf_counter())
    t0 = time.perf_counter()
    FF.preprocess_visibilities(u, v, V, w)
    tp = time.perf_counter() - t0
    d = np.diff(t) * 1e3
    print("upload %.2f ms  bin (host call returns) %.2f  finalize (waits, copies M, j) %.2f  destroy %.2f  | sum %.2f | "
          "preprocess_visibilities %.2f ms" % (d[0], d[1], d[2], d[3], d.sum(), 1e3 * tp), flush=True)

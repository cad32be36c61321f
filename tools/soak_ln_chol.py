"""Soak of the LogNormal kernel's distributed Cholesky (trailing tiles in the helpers' registers, hand-overs through the L2): the same
fit over and over -- every repetition must land on the same bits (one SHA over s and p) and take about the same time; a hand-over
that is late or lost shows as a different SHA (fallback to the pivoted LU), a failed cluster as a slow fit.
    python3 tools/soak_ln_chol.py [N] [reps] [nvis]"""
import ctypes
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from frank_amd import _lib as L  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
nvis = int(float(sys.argv[3])) if len(sys.argv) > 3 else 1_000_000
f = bench.Fitter(L, N, 0)
f.nfit = nvis
f.upload(*mock_disc_visibilities(nvis, seed=0, noise_seed=50))
h = bench.HYPER
f.bin()
H0, qmn, qmx = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 1, None, None, ctypes.byref(H0), ctypes.byref(qmn), ctypes.byref(qmx)))
shas, times = {}, []
for rep in range(reps):
    s_map, p = np.empty(N), np.empty(N)
    nit = ctypes.c_int(0)
    stats = (ctypes.c_int64 * 9)()
    t1 = time.perf_counter()
    L.check(L.lib.fh_fit_lognormal(f.ctx, None, None, 1.3, 1e-35, 1e-2, h["tol"], 400, 1e5, L.ptr(s_map), L.ptr(p), ctypes.byref(nit), None, stats, None, None))
    times.append(time.perf_counter() - t1)
    k = hashlib.sha1(s_map.tobytes() + p.tobytes()).hexdigest()[:12] + " %d passes %d Hessians" % (nit.value, stats[3])
    shas[k] = shas.get(k, 0) + 1
t = np.array(times[1:])
print("N = %d, %d fits: results %s; seconds min / median / max %.3f / %.3f / %.3f" % (N, reps, shas, t.min(), np.median(t), t.max()))
print("OK" if len(shas) == 1 and t.max() < 2.0 * np.median(t) + 0.05 else "NOT OK")

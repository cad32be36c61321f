"""Create / fit / destroy in a loop and watch device memory (hipMemGetInfo) and the host's resident set: a context, its visibility
table, its fit slots, pinned mirrors, streams and events must all go when the objects do.    python3 tools/leak_check.py [cycles]"""
import ctypes
import gc
import os
import resource
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import FixedGeometry, FrankFitter  # noqa: E402
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def free_mb():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value / 2 ** 20


def rss_mb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024


cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 40
u, v, V, w = mock_disc_visibilities(200000, seed=5, noise_seed=6)
geom = FixedGeometry(**MOCK_GEOMETRY)
log = []
for c in range(cycles):
    N = (300, 120, 400, 700)[c % 4]
    method = "LogNormal" if (c % 8 == 5) else "Normal"
    FF = FrankFitter(2.0, N, geom, verbose=False, method=method, max_iter=300 if method == "Normal" else 20,
                     convergence_failure="ignore")
    sol = FF.fit(u, v, V, w)
    assert np.all(np.isfinite(sol.I))
    del sol, FF
    gc.collect()
    log.append((free_mb(), rss_mb()))
    if c % 4 == 3:
        print("cycle %3d  device free %.1f MB  host max RSS %.1f MB" % (c, *log[-1]), flush=True)
first, last = log[7], log[-1]  # (after two rounds of the four sizes: allocator pools are warm)
print("device free: %.1f -> %.1f MB (delta %.1f), host max RSS %.1f -> %.1f MB" % (first[0], last[0], last[0] - first[0], first[1], last[1]))
leak = first[0] - last[0] > 64 or last[1] - first[1] > 256
print("LEAK" if leak else "no leak")
sys.exit(1 if leak else 0)

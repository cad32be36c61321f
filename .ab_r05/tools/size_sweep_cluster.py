"""Every form of the fit loop against the one-workgroup right-looking kernel over a range of basis sizes: bit equality of mu, p and
the iteration count.  Forms: cluster (default size), cluster of 3 and of 8, left-looking (N <= 335).
   python tools/size_sweep_cluster.py [first last step]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import _lib as L  # noqa: E402
from frank_amd import FrankFitter, FixedGeometry  # noqa: E402
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities  # noqa: E402

first, last, step = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (112, 1023, 13)))
u, v, V, w = mock_disc_visibilities(40000, seed=5, noise_seed=6)


def fit(ctx, N, M, j, max_iter):
    mu, p, nit = np.empty(N), np.empty(N), ctypes.c_int(0)
    rc = L.lib.fh_fit_normal(ctx, L.ptr(M), L.ptr(j), 1.2, 1e-15, 1e-3, 1e-3, max_iter, L.ptr(mu), L.ptr(p), ctypes.byref(nit), None, None)
    wg = ctypes.c_int(0)
    L.check(L.lib.fh_fit_cluster_info(ctx, ctypes.byref(wg), None))
    return rc, mu, p, nit.value, wg.value


bad = 0
for N in list(range(first, last + 1, step)) + [last]:
    FF = FrankFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), verbose=False)
    m = FF.preprocess_visibilities(u, v, V, w)
    M, j = np.ascontiguousarray(m["M"]), np.ascontiguousarray(m["j"])
    ctx = FF._DHT.context()
    it = 25 if N > 400 else 60
    os.environ["FRANK_AMD_K2_CLUSTER"] = "1"
    os.environ["FRANK_AMD_K2_LL"] = "0"
    ref = fit(ctx, N, M, j, it)
    forms = [("cluster default", {"FRANK_AMD_K2_CLUSTER": None}), ("cluster 3", {"FRANK_AMD_K2_CLUSTER": "3"}), ("cluster 8", {"FRANK_AMD_K2_CLUSTER": "8"})]
    if N <= 335:
        forms.append(("left-looking", {"FRANK_AMD_K2_CLUSTER": "1", "FRANK_AMD_K2_LL": "1"}))
    line = []
    for name, env in forms:
        for k_, v_ in env.items():
            if v_ is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v_
        got = fit(ctx, N, M, j, it)
        ok = got[0] == ref[0] and got[3] == ref[3] and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
        bad += not ok
        line.append("%s (%d wg): %s" % (name, got[4], "equal" if ok else "DIFFERENT rc %d/%d it %d/%d" % (got[0], ref[0], got[3], ref[3])))
        os.environ["FRANK_AMD_K2_LL"] = "0"
    print("N=%4d  rc %d, %d iterations | %s" % (N, ref[0], ref[3], " | ".join(line)), flush=True)
print("sizes with a difference:", bad)

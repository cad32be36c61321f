"""The persistent fit loop against the library loop (rocBLAS + rocSOLVER per pass) beyond N = 320: us per pass, on one
workgroup and on a cluster.   python tools/k2_wide.py [sizes...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import FrankFitter, FixedGeometry  # noqa: E402
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [340, 400, 511, 639, 640, 700, 800, 900, 1000, 1023]
u, v, V, w = mock_disc_visibilities(60000, seed=31, noise_seed=32)
for N in sizes:
    kw = dict(alpha=1.3, weights_smooth=1e-2, verbose=False, store_iteration_diagnostics=True, max_iter=150, convergence_failure="ignore")
    res = {}
    pre = None
    for mode, env in (("one workgroup", {"FRANK_AMD_K2_CLUSTER": "1"}), ("cluster", {}), ("library", {"FRANK_AMD_K2": "rocsolver"})):
        for k_, v_ in env.items():
            os.environ[k_] = v_
        try:
            FF = FrankFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), **kw)
            if pre is None:
                pre = FF.preprocess_visibilities(u, v, V, w)
            m = dict(pre, hash=[False, FF._DHT, FF._geometry, "opt_thick", None])
            FF.fit_preprocessed(m)
            t0 = time.perf_counter()
            sol = FF.fit_preprocessed(m)
            dt = time.perf_counter() - t0
            res[mode] = (1e6 * dt / (FF.iteration_diagnostics["num_iterations"] + 2), sol.I.copy())
        finally:
            for k_ in env:
                del os.environ[k_]
    ref = res["library"][1]
    print("N=%4d  us per pass: one workgroup %.0f  cluster %.0f  library loop %.0f   max|dI|/max|I| vs library: %.1e %.1e" % (
        N, res["one workgroup"][0], res["cluster"][0], res["library"][0],
        np.abs(res["one workgroup"][1] - ref).max() / np.abs(ref).max(), np.abs(res["cluster"][1] - ref).max() / np.abs(ref).max()), flush=True)

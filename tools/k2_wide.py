"""Fit loop beyond N = 320: the one-panel persistent kernel against the library loop (FRANK_AMD_K2=rocsolver), microseconds per
power-spectrum iteration.   python3 tools/k2_wide.py"""
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from frank_amd import FixedGeometry, FrankFitter
    from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities
    u, v, V, w = mock_disc_visibilities(100000, seed=31, noise_seed=32)
    for N in (300, 320, 340, 400, 478, 511, 600, 639):
        FF = FrankFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), alpha=1.3, weights_smooth=1e-2, verbose=False,
                         store_iteration_diagnostics=True)
        pre = FF.preprocess_visibilities(u, v, V, w)
        FF.fit_preprocessed(pre)
        t0 = time.perf_counter()
        FF.fit_preprocessed(pre)
        dt = time.perf_counter() - t0
        nit = FF.iteration_diagnostics["num_iterations"]
        print("  N=%d  %d iterations  %.1f ms  %.1f us per iteration" % (N, nit, 1e3 * dt, 1e6 * dt / max(nit, 1)))
else:
    for mode in ("persistent", "rocsolver"):
        env = dict(os.environ)
        if mode == "rocsolver":
            env["FRANK_AMD_K2"] = "rocsolver"
        print(mode)
        sys.stdout.flush()
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, timeout=600)

"""method='LogNormal' beyond the persistent kernel (320 < N <= 1023, lognormal_wide.hip): whole fits from the mock disc's
visibilities, time and work counters.   python tools/ln_wide_time.py [N ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import FixedGeometry, FrankFitter  # noqa: E402
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities  # noqa: E402

u, v, V, w = mock_disc_visibilities(1_000_000, seed=3, noise_seed=4)
for N in [int(a) for a in sys.argv[1:]] or [300, 330, 400, 640, 1000]:
    FF = FrankFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), method="LogNormal", alpha=1.3, weights_smooth=1e-2, max_iter=200,
                     verbose=False, check_qbounds=False, convergence_failure="ignore", store_iteration_diagnostics=True)
    t0 = time.perf_counter()
    sol = FF.fit(u, v, V, w)
    dt = time.perf_counter() - t0
    st = getattr(getattr(sol, "_fit", None), "_newton_stats", None) or getattr(getattr(FF, "_sol", None), "_fit", None) and FF._sol._fit._newton_stats
    print("N = %4d: %.2f s, %d passes, I in [%.3g, %.3g]%s" % (
        N, dt, FF.iteration_diagnostics["num_iterations"], sol.I.min(), sol.I.max(),
        "" if st is None else "  (MAP solves %d, Newton steps %d, evaluations %d, Hessians %d)" % tuple(st[:4])), flush=True)

#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: 512 fits of one mapping (alpha x w_smooth grid), N=300, 1e6 visibilities.

Reference semantics: fit.py:534-548 (`run_multiple_fits`) -- one full fit per grid point; here the mapping is
binned once and the 512 power-spectrum iterations run as one batched fit_loop launch (one CU per fit).
Prints one JSON line; not the headline bench (bench.py is)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import FixedGeometry, FrankFitter  # noqa: E402
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities  # noqa: E402
from frank_amd.sweep import sweep_fits  # noqa: E402

N, NVIS = 300, int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 6
alphas, wss = np.meshgrid(np.linspace(1.01, 1.5, 32), np.logspace(-4, -1, 16))
alphas, wss = alphas.ravel(), wss.ravel()
u, v, V, w = mock_disc_visibilities(NVIS, seed=0, noise_seed=50)
FF = FrankFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), verbose=False)
FF.preprocess_visibilities(u[:1000], v[:1000], V[:1000], w[:1000])  # warm-up (context, kernels)
from frank_amd import sweep as _sweep  # noqa: E402
svd = {"points": 0, "s": 0.0}
_plain = _sweep._refit_through_svd_route


def _timed_refit(*a, **k):  # points whose device loop met a failed Cholesky continue one posterior at a time, as the reference
    t = time.perf_counter()
    out = _plain(*a, **k)
    svd["points"] += 1
    svd["s"] += time.perf_counter() - t
    return out


_sweep._refit_through_svd_route = _timed_refit
# warm-up of the batched launch too (code objects, rocBLAS kernels, slot buffers: seconds on a box that has just come up)
sweep_fits(FF, FF.preprocess_visibilities(u[:1000], v[:1000], V[:1000], w[:1000]), alphas[:2], wss[:2], max_iter=5)
svd["points"], svd["s"] = 0, 0.0
t0 = time.perf_counter()
m = FF.preprocess_visibilities(u, v, V, w)
t1 = time.perf_counter()
sols, niters = sweep_fits(FF, m, alphas, wss, max_iter=2000)
t2 = time.perf_counter()
# spot-check two points against the one-at-a-time path
for k in (0, 511):
    F1 = FrankFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), alpha=alphas[k], weights_smooth=wss[k], verbose=False,
                     convergence_failure="ignore")
    assert np.array_equal(F1.fit_preprocessed(m).I, sols[k].I)
print(json.dumps({"config": "512 fits, alpha in linspace(1.01,1.5,32) x w_smooth in logspace(-4,-1,16), N=300, "
                            "%d visibilities, shared (M, j)" % NVIS,
                  "map_s": t1 - t0, "sweep_s": t2 - t1, "fits_per_s": alphas.size / (t2 - t0),
                  "points_through_the_svd_route": svd["points"], "svd_route_s": svd["s"],
                  "fits_per_s_of_the_batched_launch": (alphas.size - svd["points"]) / (t2 - t1 - svd["s"]),
                  "iterations_min_med_max": [int(np.min(niters)), int(np.median(niters)), int(np.max(niters))],
                  "not_converged": int(np.sum(np.array(niters) >= 2000))}))

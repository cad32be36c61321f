"""BASELINE configs[2] at fp64: N=300, 1e7 visibilities, method='LogNormal' (alpha=1.3, w_smooth=1e-2 as in the reference's
own LogNormal test, tests.py:350) end to end on one GPU (development tool; prints one JSON line)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import FrankFitter, FixedGeometry
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 7
u, v, V, w = mock_disc_visibilities(n, seed=0, noise_seed=50)
FF = FrankFitter(2.0, 300, FixedGeometry(**MOCK_GEOMETRY), alpha=1.3, weights_smooth=1e-2, method="LogNormal",
                 store_iteration_diagnostics=True, verbose=False, convergence_failure="ignore")
t0 = time.perf_counter()
pre = FF.preprocess_visibilities(u, v, V, w)
t1 = time.perf_counter()
sol = FF.fit_preprocessed(pre)
t2 = time.perf_counter()
st = sol._fit._newton_stats
print(json.dumps({"config": "N=300, %d visibilities, LogNormal, alpha=1.3, w_smooth=1e-2, fp64" % n,
                  "map_s_incl_upload": t1 - t0, "fit_s": t2 - t1, "power_spectrum_iterations": FF.iteration_diagnostics["num_iterations"],
                  "map_solves": st[0], "newton_steps": st[1], "function_evaluations": st[2], "hessians": st[3],
                  "newton_exits_0_1_2_3": list(st[4:8]), "I_min": float(sol.I.min()), "I_max": float(sol.I.max())}))

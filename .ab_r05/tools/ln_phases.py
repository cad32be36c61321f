"""Where a method='LogNormal' fit at N = 300 spends its time (timing build: make -C frank_amd/csrc timing; the library prints its
phase counters to stderr after every fit): both line-search modes, one workgroup and the cluster of eight.
    FRANK_AMD_LIB=$PWD/frank_amd/libfrank_hip_timing.so python3 tools/ln_phases.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import FixedGeometry, FrankFitter  # noqa: E402
from frank_amd.mock import MOCK_GEOMETRY  # noqa: E402

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
g = np.load(os.path.join(G, "lognormal_N300_1e7.npz"))
src = np.load(os.path.join(G, str(g["source"])))
for cluster in ("1", None):
    if cluster is None:
        os.environ.pop("FRANK_AMD_LN_CLUSTER", None)
    else:
        os.environ["FRANK_AMD_LN_CLUSTER"] = cluster
    for ls in ("linear", "reference"):
        FF = FrankFitter(2.0, 300, FixedGeometry(**MOCK_GEOMETRY), alpha=float(g["alpha"]), weights_smooth=float(g["wsmooth"]), method="LogNormal",
                         I_scale=float(g["I_scale"]), store_iteration_diagnostics=True, verbose=False, check_qbounds=False,
                         convergence_failure="ignore", lognormal_linesearch=ls)
        m = {"M": src["M"], "j": src["j"], "null_likelihood": 0.0, "hash": [False, FF._DHT, FF._geometry, "opt_thick", None]}
        sys.stderr.write("--- %s, line search '%s'\n" % ("one workgroup" if cluster else "cluster (default)", ls))
        sys.stderr.flush()
        t = time.time()
        sol = FF.fit_preprocessed(m)
        dt = time.time() - t
        st = sol._fit._newton_stats
        sys.stderr.write("    %.3f s, %d power-spectrum iterations, Newton steps %d, evaluations %d, Hessian factorisations %s\n" % (
            dt, FF.iteration_diagnostics["num_iterations"], st[1], st[2], st[3] if len(st) > 3 else "?"))
        sys.stderr.flush()

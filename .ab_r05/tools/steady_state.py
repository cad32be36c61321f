"""Steady-state throughput of the fit pipeline (binning pass of fit i+1 beside the iterations of the fits before it):
fits/s over a run long enough that the one drain at its end does not matter.

    python3 tools/steady_state.py [steps] [--distinct R]     # R > 0: a ring of R resident tables, range cache off
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from frank_amd import _lib as L  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 2000
ring = int(sys.argv[sys.argv.index("--distinct") + 1]) if "--distinct" in sys.argv else 0
f = bench.Fitter(L, 300, 0)
if os.environ.get("BIN_CUS"):
    L.check(L.lib.fh_ctx_set_cu_partition(f.ctx, int(os.environ["BIN_CUS"])))
nvis = int(float(os.environ.get("NVIS", "1e7")))  # (development: a smaller table, to separate the cost of the binning traffic)
f.nfit = nvis
f.upload(*mock_disc_visibilities(nvis, seed=0, noise_seed=50))
f.fit()
print("slots", os.environ.get("FRANK_AMD_FIT_SLOTS", "default"), "bin_cus", os.environ.get("BIN_CUS"), bench.steady_state(f, L, steps, ring))

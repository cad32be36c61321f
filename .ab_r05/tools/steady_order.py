"""Is the gap between bench.py's two steady-state legs the form of the loop or the order of the legs?  One process, one context:
the same 2 s window by load, by load again, forced register-resident, by load.   python3 tools/steady_order.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from frank_amd import _lib as L  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

f = bench.Fitter(L, 300, 0)
f.nfit = 10_000_000
f.upload(*mock_disc_visibilities(f.nfit, seed=0, noise_seed=50))
f.fit()
for tag, env in (("by load", None), ("by load", None), ("forced", "1"), ("by load", None), ("in memory", "0"), ("by load", None)):
    if env is None:
        os.environ.pop("FRANK_AMD_K2_RR", None)
    else:
        os.environ["FRANK_AMD_K2_RR"] = env
    r = bench.steady_state(f, L)
    print("%-10s %.0f fits/s (%d steps in %.2f s)" % (tag, r["fits_per_s"], r["steps"], r["seconds"]), flush=True)

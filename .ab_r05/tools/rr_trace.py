"""Per-wave timeline of ONE pass of the register-resident fit loop (timing build: make -C frank_amd/csrc timing).
    FRANK_AMD_LIB=$PWD/frank_amd/libfrank_hip_timing.so FRANK_AMD_K2_RR=1 python3 tools/rr_trace.py
"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["FRANK_AMD_K2_CLUSTER"] = "1"
os.environ.setdefault("FRANK_AMD_K2_RR", "1")
from frank_amd import FixedGeometry, FrankFitter, _lib  # noqa: E402
from frank_amd.mock import MOCK_GEOMETRY  # noqa: E402

g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "fit_N300_1e6.npz"))
FF = FrankFitter(2.0, 300, FixedGeometry(**MOCK_GEOMETRY), verbose=False, store_iteration_diagnostics=True)
m = {"M": g["M"], "j": g["j"], "null_likelihood": 0.0, "hash": [False, FF._DHT, FF._geometry, "opt_thick", None]}
FF.fit_preprocessed(m)
t = time.time()
FF.fit_preprocessed(m)
dt = time.time() - t
nit = FF.iteration_diagnostics["num_iterations"]
print("%d passes, %.1f us per pass" % (nit, 1e6 * dt / nit))
nw = 8  # (wave 7: the chain specialist)
tr = (ctypes.c_longlong * 2048)()
_lib.lib.fh_debug_loop_trace(FF._DHT.context(), tr)
t = np.array(tr[: nw * 120], dtype=np.int64).reshape(nw, 20, 6)
t0 = t[:, :, 0][t[:, :, 0] > 0].min()
us = lambda v: (v - t0) / 2.4e3
print("prologue (entry of the solve to the start of step 0), per wave: " + " ".join("%.2f" % ((t[w, 0, 0] - t[w, 0, 5]) / 2.4e3) for w in range(nw)) + " us")
last = t[:, 18, :][t[:, 18, :] > 0].max()
print("entry to the end of step 18: %.1f us" % ((last - t[:, 0, 5][t[:, 0, 5] > 0].min()) / 2.4e3))
print("step: start | per wave: [chain done] pass done, flag seen, columns done   (us after the start of the step; * = the chain wave)")
for k in range(19):
    row = "%2d %7.2f |" % (k, us(t[:, k, 0][t[:, k, 0] > 0].min()))
    for w in range(nw):
        r = t[w, k, :]
        f = lambda v: (us(v) - us(r[0])) if v > 0 else float("nan")
        row += " %s%5.2f %5.2f %5.2f |" % (("*%4.1f " % f(r[1])) if r[1] > 0 else "", f(r[2]), f(r[3]), f(r[4]))
    print(row)

"""Host-side timeline of one pipelined step of bench.py (where do the milliseconds between two bin_gram kernels go?).
    python tools/step_timeline.py [steps]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import bench  # noqa: E402
from frank_amd import _lib as L  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
f = bench.Fitter(L, bench.N_COLL, 0)
f.nfit = bench.N_VIS
f.upload(*mock_disc_visibilities(bench.N_VIS, seed=0, noise_seed=50))
f.run_steps(3)
f.sync()
h = bench.HYPER
H0, qmn, qmx = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
slots = L.lib.fh_fit_slots()
pending = []
rows = []
t_start = time.perf_counter()
for i in range(steps):
    t0 = time.perf_counter()
    if len(pending) == slots:
        f.collect(pending.pop(0))
    t1 = time.perf_counter()
    L.check(L.lib.fh_bin_reset(f.ctx))
    L.check(L.lib.fh_bin_visibilities(f.ctx, ctypes.byref(f.geom), f.vis, 0, f.nfit))
    t2 = time.perf_counter()
    L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 0, None, None, None, None, None))
    t3 = time.perf_counter()
    t = ctypes.c_int(-1)
    L.check(L.lib.fh_fit_submit(f.ctx, h["alpha"], h["p0"], h["wsmooth"], h["tol"], h["max_iter"], ctypes.byref(t)))
    t4 = time.perf_counter()
    pending.append(t.value)
    rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, f.kernel_ms()))
L.check(L.lib.fh_fit_flush(f.ctx))
for t in pending:
    f.collect(t)
f.sync()
total = time.perf_counter() - t_start
a = np.array(rows[5:])
print("slots", slots, "steps", steps, "total %.1f ms/step" % (1e3 * total / steps))
print("mean ms per step: collect %.3f  bin_visibilities (host call) %.3f  stats_finalize %.3f  fit_submit %.3f | bin_gram kernel "
      "(previous step's, by events) %.3f" % tuple(list(1e3 * a[:, :4].mean(0)) + [a[:, 4].mean()]))
pm = ctypes.c_float(0)
L.check(L.lib.fh_bin_last_prepass_ms(f.ctx, ctypes.byref(pm)))
print("last pre-pass (deproject .. sort, by events): %.3f ms" % pm.value)

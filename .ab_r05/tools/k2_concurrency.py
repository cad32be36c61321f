"""Duration of n concurrent fit loops (one launch, or the early launches of a filling pipeline) against n, on one compute unit
each and on clusters: how much a fit loop slows down when others share its XCD's L2.
   FRANK_AMD_K2_CLUSTER=1|5 FRANK_AMD_FIT_EARLY=0|1 python tools/k2_concurrency.py [n ...]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import bench  # noqa: E402
from frank_amd import _lib as L  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

f = bench.Fitter(L, bench.N_COLL, 0)
f.nfit = 1_000_000
f.upload(*mock_disc_visibilities(f.nfit, seed=0, noise_seed=50))
f.fit()
h = bench.HYPER
ns = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16, 32, 64, 128]
for n in ns:
    f.bin()
    L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 0, None, None, None, None, None))
    f.sync()
    t0 = time.perf_counter()
    tickets = []
    for i in range(n):
        t = ctypes.c_int(-1)
        L.check(L.lib.fh_fit_submit(f.ctx, h["alpha"], h["p0"], h["wsmooth"], h["tol"], h["max_iter"], ctypes.byref(t)))
        tickets.append(t.value)
    L.check(L.lib.fh_fit_flush(f.ctx))
    for t in tickets:
        nit = f.collect(t)
    dt = time.perf_counter() - t0
    print("%4d concurrent fits: %.1f ms (%d iterations, %.1f us per iteration; cluster fall-backs so far %d)" % (
        n, 1e3 * dt, nit, 1e6 * dt / nit, f.cluster_info()[1]), flush=True)

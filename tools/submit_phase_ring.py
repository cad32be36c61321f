"""The driver's K-step region on the ring of tables with the range cache off, taken apart: host time of the K submissions, end of the
region; with and without the range look-ahead (fh_bin_prefetch_range), against the region on one table with the caches on.
   python tools/submit_phase_ring.py [K]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from frank_amd import _lib as L  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
f = bench.Fitter(L, 300, 0)
for k in range(4):
    u, v, V, w = mock_disc_visibilities(10_000_000, seed=7000 + k, noise_seed=7100 + k)
    f.upload(u, v, V, w)
f.vis = f.tables[0]
for _ in range(2):
    f.fit()
f.run_steps(5, ring=f.tables)


def region(tag, ring, look):
    f.sync()
    t0 = time.perf_counter()
    tickets = []
    if ring and look:
        L.check(L.lib.fh_bin_prefetch_range(f.ctx, ctypes.byref(f.geom), ring[0], 0, f.nfit))
    for i in range(K):
        if ring and look and i + 1 < K:
            L.check(L.lib.fh_bin_prefetch_range(f.ctx, ctypes.byref(f.geom), ring[(i + 1) % len(ring)], 0, f.nfit))
        tickets.append(f.submit(None if not ring else ring[i % len(ring)]))
    t1 = time.perf_counter()
    f.sync()
    t1b = time.perf_counter()
    L.check(L.lib.fh_fit_flush(f.ctx))
    nit = [f.collect(t) for t in tickets]
    t2 = time.perf_counter()
    print("%-46s submissions %.2f ms (%.3f each), binning stream idle at %.2f ms, region %.2f ms -> %.1f fits/s (%d passes)" % (
        tag, 1e3 * (t1 - t0), 1e3 * (t1 - t0) / K, 1e3 * (t1b - t0), 1e3 * (t2 - t0), K / (t2 - t0), nit[-1]))


for rep in range(2):
    region("one table, caches on", None, False)
    L.check(L.lib.fh_ctx_set_range_cache(f.ctx, 0))
    region("ring of 4, range cache off, no look-ahead", f.tables, False)
    region("ring of 4, range cache off, look-ahead", f.tables, True)
    L.check(L.lib.fh_ctx_set_range_cache(f.ctx, 1))

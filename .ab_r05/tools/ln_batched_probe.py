"""bench.py's extra.lognormal_batched64 workload with the iteration counts laid out on the (w_smooth, alpha) grid, with and without
the lead point on a cluster (FRANK_AMD_LN_SWEEP_CLUSTERS)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from frank_amd import _lib as L
from frank_amd.mock import mock_disc_visibilities
N = 300
f = bench.Fitter(L, N, 0)
f.nfit = 1_000_000
f.upload(*mock_disc_visibilities(10_000_000, seed=0, noise_seed=50))
f.fit()
h = bench.HYPER
al, ws = np.meshgrid(np.linspace(1.2, 1.5, 8), np.logspace(-3, -1, 8))
al, ws = np.ascontiguousarray(al.ravel()), np.ascontiguousarray(ws.ravel())
B = al.size
p0 = np.full(B, 1e-35)
L.check(L.lib.fh_ctx_set_lognormal_linesearch(f.ctx, 0))
for env in ("0", "1"):
    os.environ["FRANK_AMD_LN_SWEEP_CLUSTERS"] = env
    s_map, pp = np.empty((B, N)), np.empty((B, N))
    niter, status = (ctypes.c_int * B)(), (ctypes.c_int * B)()
    stats = (ctypes.c_int64 * (9 * B))()
    t0 = time.perf_counter()
    f.bin(1_000_000)
    L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 0, None, None, None, None, None))
    L.check(L.lib.fh_fit_lognormal_batched(f.ctx, None, None, B, L.ptr(al), L.ptr(p0), L.ptr(ws), h["tol"], h["max_iter"], 1e5, L.ptr(s_map), L.ptr(pp),
                                           niter, status, stats))
    dt = time.perf_counter() - t0
    its = np.array(list(niter)).reshape(8, 8)
    print("LN_SWEEP_CLUSTERS=%s: %.2f s, %.1f fits/s; sha %s" % (env, dt, B / dt, __import__("hashlib").sha256(s_map.tobytes() + pp.tobytes()).hexdigest()[:12]))
    if env == "0":
        print("iterations, rows = w_smooth 1e-3 .. 1e-1, columns = alpha 1.2 .. 1.5:")
        print(its)

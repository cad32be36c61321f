"""BASELINE configs[4], distinct-datasets variant: 512 fits over the alpha x w_smooth grid (32 x 16), every fit binning
its OWN 1e6-visibility table (8 resident tables with seeds 0..7, cycled -- generating 512 on the host would take half
an hour and the device work is the same), through the batched pipeline (development tool; prints one JSON line)."""
import ctypes, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from frank_amd import _lib as L
from frank_amd.mock import mock_disc_visibilities

NT, NV = 8, 10 ** 6
fit = [bench.Fitter(L, 300, 0)]
f = fit[0]
tables = []
for s in range(NT):
    u, v, V, w = mock_disc_visibilities(NV, seed=s, noise_seed=50 + s)
    f.upload(u, v, V, w)
    tables.append((f.vis, f.n))
alphas = np.linspace(1.01, 1.5, 32)
wss = np.logspace(-4, -1, 16)
grid = [(a, ws) for a in alphas for ws in wss]
slots = L.lib.fh_fit_slots()
mu, p, nit = np.empty(300), np.empty(300), ctypes.c_int()

def run(points):
    pend, its = [], []
    for i, (a, ws) in enumerate(points):
        if len(pend) == slots:
            L.check(L.lib.fh_fit_collect(f.ctx, pend.pop(0), L.ptr(mu), L.ptr(p), ctypes.byref(nit))); its.append(nit.value)
        f.vis, f.n = tables[i % NT]
        f.bin()
        H0, q0, q1 = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 1, None, None, ctypes.byref(H0), ctypes.byref(q0), ctypes.byref(q1)))
        t = ctypes.c_int(-1)
        L.check(L.lib.fh_fit_submit(f.ctx, a, 1e-15, ws, 1e-3, 2000, ctypes.byref(t)))
        pend.append(t.value)
    L.check(L.lib.fh_fit_flush(f.ctx))
    for t in pend:
        L.check(L.lib.fh_fit_collect(f.ctx, t, L.ptr(mu), L.ptr(p), ctypes.byref(nit))); its.append(nit.value)
    return its

run(grid[:32]); f.sync()
t0 = time.perf_counter()
its = run(grid); f.sync()
dt = time.perf_counter() - t0
print(json.dumps({"config": "512 fits, alpha in linspace(1.01,1.5,32) x w_smooth in logspace(-4,-1,16), each binning its own "
                            "1e6-visibility table (8 resident tables cycled), N=300, fp64, one GPU", "seconds": dt,
                  "fits_per_s": len(grid) / dt, "iterations_min_median_max": [int(np.min(its)), int(np.median(its)), int(np.max(its))]}))

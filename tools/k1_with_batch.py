"""bin_gram kernel time while ONE launch of B fit loops (B workgroups dealt evenly over the XCDs / shader engines by a
single dispatch) runs beside it, against the same number started one by one (development tool)."""
import ctypes, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from frank_amd import _lib as L
from frank_amd.mock import mock_disc_visibilities

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
f = bench.Fitter(L, 300, 0)
u, v, V, w = mock_disc_visibilities(10 ** 7, seed=0, noise_seed=50)
f.upload(u, v, V, w)
f.fit(); f.sync()
# normal equations of this table for the batched launches (second context: its own stream)
N = 300
M, j = np.empty((N, N)), np.empty(N)
H0, a, b = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 1, L.ptr(M), L.ptr(j), ctypes.byref(H0), ctypes.byref(a), ctypes.byref(b)))
g = bench.Fitter(L, 300, 0)
alphas, p0, ws = np.full(B, 1.05), np.full(B, 1e-15), np.full(B, 1e-4)
mu, p = np.empty((B, N)), np.empty((B, N))
niter, status = (ctypes.c_int * B)(), (ctypes.c_int * B)()
stop = False
def sweeps():
    while not stop:
        L.check(L.lib.fh_fit_normal_batched(g.ctx, L.ptr(M), L.ptr(j), B, L.ptr(alphas), L.ptr(p0), L.ptr(ws), 1e-3, 2000,
                                            L.ptr(mu), L.ptr(p), niter, status))
t = threading.Thread(target=sweeps); t.start()
time.sleep(0.5)
ks = []
for i in range(40):
    f.bin(); ks.append(f.kernel_ms())
stop = True; t.join()
print("bin_gram beside ONE launch of %d fit loops (niter %d): " % (B, niter[0]) + " ".join("%.1f" % k for k in ks))
print("median %.2f ms" % np.median(ks))

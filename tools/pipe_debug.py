import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from frank_amd import _lib as L
from frank_amd.mock import mock_disc_visibilities
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
f = bench.Fitter(L, 300, 0)
u, v, V, w = mock_disc_visibilities(n, seed=0, noise_seed=50)
f.upload(u, v, V, w)
nit = f.fit(); mu0 = f.mu.copy(); print("sync fit niter", nit)
for trial in range(3):
    ts = []
    for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
        try:
            ts.append(f.submit())
        except RuntimeError as e:
            print("submit failed", e); break
    for t in ts:
        try:
            k = f.collect(t); print(" trial", trial, "ticket", t, "niter", k, "max|dmu|/max", np.abs(f.mu - mu0).max() / np.abs(mu0).max())
        except RuntimeError as e:
            print(" collect failed", t, e)

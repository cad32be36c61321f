"""bin_gram kernel time over back-to-back launches WITHOUT any fit loop in flight, then with (development tool):
separates clock/power drift under sustained fp64 load from interference by co-running fit_loop kernels."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from frank_amd import _lib as L
from frank_amd.mock import mock_disc_visibilities
f = bench.Fitter(L, 300, 0)
u, v, V, w = mock_disc_visibilities(10 ** 7, seed=0, noise_seed=50)
f.upload(u, v, V, w)
f.run_steps(2); f.sync()
time.sleep(1.0)
ks = []
t0 = time.perf_counter()
for i in range(40):
    f.bin()
    ks.append(f.kernel_ms())
f.sync()
print("alone, 40 back-to-back (%.0f ms wall): " % (1e3 * (time.perf_counter() - t0)) + " ".join("%.1f" % k for k in ks))
time.sleep(1.0)
ks = []
f.run_steps(40, ks); f.sync()
print("pipelined with fit loops, 40 steps: " + " ".join("%.1f" % k for k in ks))

"""bin_gram kernel time of the first pipelined steps vs the number of fit_loop kernels in flight (development tool)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from frank_amd import _lib as L
from frank_amd.mock import mock_disc_visibilities
f = bench.Fitter(L, 300, 0)
u, v, V, w = mock_disc_visibilities(10 ** 7, seed=0, noise_seed=50)
f.upload(u, v, V, w)
f.run_steps(2); f.sync()
pend = []
t0 = time.perf_counter()
for i in range(24):
    if len(pend) == L.lib.fh_fit_slots():
        f.collect(pend.pop(0))
    pend.append(f.submit())
    print("step %2d  t=%6.1f ms  kernel %.2f ms  fits in flight (0.26 s each) ~%d" % (
        i, 1e3 * (time.perf_counter() - t0), f.kernel_ms(), min(i, int(0.26 / 0.0345))))
for t in pend:
    f.collect(t)

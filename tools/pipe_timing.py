"""Host-side timing of the phases of one pipelined step (development tool)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import bench
from frank_amd import _lib as L
from frank_amd.mock import mock_disc_visibilities
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 7
K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
f = bench.Fitter(L, 300, 0)
u, v, V, w = mock_disc_visibilities(n, seed=0, noise_seed=50)
f.upload(u, v, V, w)
f.run_steps(3)
f.sync()
lib = L.lib
H0, qmn, qmx = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
h = bench.HYPER
acc = np.zeros(5)
pending = []
t_all = time.perf_counter()
for i in range(K):
    t0 = time.perf_counter()
    if len(pending) == lib.fh_fit_slots():
        f.collect(pending.pop(0))
    t1 = time.perf_counter()
    L.check(lib.fh_bin_reset(f.ctx))
    L.check(lib.fh_bin_visibilities(f.ctx, ctypes.byref(f.geom), f.vis, 0, f.n))
    t2 = time.perf_counter()
    L.check(lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 1, None, None, ctypes.byref(H0), ctypes.byref(qmn),
                                  ctypes.byref(qmx)))
    t3 = time.perf_counter()
    t = ctypes.c_int(-1)
    L.check(lib.fh_fit_submit(f.ctx, h["alpha"], h["p0"], h["wsmooth"], h["tol"], h["max_iter"], ctypes.byref(t)))
    t4 = time.perf_counter()
    pending.append(t.value)
    ms = f.kernel_ms()
    t5 = time.perf_counter()
    acc += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4]
    if i % 10 == 9:
        print("step %3d: collect %.2f  bin(launch) %.2f  finalize(sync) %.2f  submit %.2f  kernel_ms() %.2f | kernel %.1f ms"
              % (i, *(1e3 * acc / 10), ms))
        acc[:] = 0
for t in pending:
    f.collect(t)
print("total %.1f ms/step" % (1e3 * (time.perf_counter() - t_all) / K))

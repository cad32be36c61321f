"""bin_gram (fused, register-resident) vs rows-to-memory + rocBLAS dsyrk at the same N (development tool)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from frank_amd import _lib as L
from frank_amd.mock import mock_disc_visibilities
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 7
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
f = bench.Fitter(L, N, 0)
u, v, V, w = mock_disc_visibilities(n, seed=0, noise_seed=50)
f.upload(u, v, V, w)
for rep in range(4):
    t = time.perf_counter()
    f.bin()
    f.sync()
    dt = time.perf_counter() - t
    print("%s N=%d n=%d: bin pass %.2f ms (events: %.2f ms)" % (os.environ.get("FRANK_AMD_K1", "fused"), N, n, dt * 1e3,
                                                                f.kernel_ms()))
M, j = np.empty((N, N)), np.empty(N)
H0, a, b = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 1, L.ptr(M), L.ptr(j), ctypes.byref(H0), ctypes.byref(a),
                                ctypes.byref(b)))
np.save("gpurun_out/M_%s.npy" % os.environ.get("FRANK_AMD_K1", "fused"), M)
print("   H0 %.15e  |M|_F %.15e" % (H0.value, np.linalg.norm(M)))

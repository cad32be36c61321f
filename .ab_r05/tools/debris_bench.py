"""vis_model='debris' at N=300, 1e7 visibilities: bin_gram pass time (development tool)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from frank_amd import _lib as L
from frank_amd.mock import mock_disc_visibilities
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 7
N = 300
f = bench.Fitter(L, N, 0)
r = np.empty(N)
L.check(L.lib.fh_dht_get(f.dht, L.ptr(r), None, None, None, None, None, None))
from frank_amd.constants import rad_to_arcsec
H = 0.02 + 0.05 * r * rad_to_arcsec
H2 = 0.5 * (2 * np.pi * H / rad_to_arcsec) ** 2
L.check(L.lib.fh_ctx_set_scale_height(f.ctx, L.ptr(np.ascontiguousarray(H2))))
u, v, V, w = mock_disc_visibilities(n, seed=0, noise_seed=50)
f.upload(u, v, V, w)
for rep in range(3):
    t = time.perf_counter(); f.bin(); f.sync(); dt = time.perf_counter() - t
    print("%s debris N=%d n=%d: bin pass %.2f ms (events: %.2f ms)" % (os.environ.get("FRANK_AMD_K1", "fused"), N, n, dt * 1e3, f.kernel_ms()))

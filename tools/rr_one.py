"""One fit of the register-resident loop at size N against the other forms: status, passes, time."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["FRANK_AMD_K2_CLUSTER"] = "1"
from frank_amd import FixedGeometry, FrankFitter, _lib
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities
u, v, V, w = mock_disc_visibilities(100000, seed=31, noise_seed=32)
for N in [int(a) for a in sys.argv[1:]]:
    FF = FrankFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), verbose=False)
    m = FF.preprocess_visibilities(u, v, V, w)
    M, j = np.ascontiguousarray(m["M"]), np.ascontiguousarray(m["j"])
    ctx = FF._DHT.context()
    for d in ("0", "1"):
        os.environ["FRANK_AMD_K2_RR"] = d
        mu, p, nit = np.zeros(N), np.zeros(N), ctypes.c_int(0)
        t = time.time()
        rc = _lib.lib.fh_fit_normal(ctx, _lib.ptr(M), _lib.ptr(j), 1.05, 1e-15, 1e-4, 1e-3, 80, _lib.ptr(mu), _lib.ptr(p), ctypes.byref(nit), None, None)
        print("N %d RR=%s rc %d passes %d  %.1f ms  mu[:3] %s" % (N, d, rc, nit.value, 1e3 * (time.time() - t), mu[:3]), flush=True)
    print("max|T_(6,k+1)| max|X| slot at steps 0..4:", p[:20].reshape(5, 4))

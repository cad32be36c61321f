import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from frank_amd import FrankFitter, FixedGeometry
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities
n = 10**7
u, v, V, w = mock_disc_visibilities(n, seed=0, noise_seed=50)
FF = FrankFitter(2.0, 300, FixedGeometry(**MOCK_GEOMETRY), verbose=False)
sol = FF.fit(u, v, V, w)
for rep in range(3):
    t0 = time.perf_counter(); P = sol.predict(u, v); t1 = time.perf_counter()
    chi2 = np.sum(w * np.abs(P - V) ** 2); t2 = time.perf_counter()
    print("predict(u, v) at 1e7: %.3f s; chi2 on the host %.3f s -> %.6e" % (t1 - t0, t2 - t1, chi2), flush=True)

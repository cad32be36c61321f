"""What a user of the reference runs, end to end, from host arrays: geometry fit (non-parametric), then the frank fit under the fitted
geometry -- FrankFitter(Rmax, N, FitGeometryFourierBessel(Rmax, 20)).fit(u, v, V, w) -- with the host-side pieces timed.
    python3 tools/end_to_end.py [nvis [N]]"""
import os
import sys
import time


sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import FrankFitter  # noqa: E402
from frank_amd.geometry import FitGeometryFourierBessel, FitGeometryGaussian  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 7
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
u, v, V, w = mock_disc_visibilities(n, seed=0, noise_seed=50)
for make in (lambda: FitGeometryFourierBessel(2.0, 20, guess=[30.0, 80.0, 0.0, 0.0]), lambda: FitGeometryGaussian(guess=[30.0, 80.0, 0.0, 0.0])):
    for rep in range(2):
        geom = make()
        t0 = time.perf_counter()
        FF = FrankFitter(2.0, N, geom, verbose=False, store_iteration_diagnostics=True)
        t1 = time.perf_counter()
        geom.fit(u, v, V, w)
        t2 = time.perf_counter()
        m = FF.preprocess_visibilities(u, v, V, w)
        t3 = time.perf_counter()
        sol = FF.fit_preprocessed(m)
        t4 = time.perf_counter()
        print("%s n=%d N=%d: fitter %.3f s, geometry fit %.3f s (inc %.3f PA %.3f), mapping from host arrays %.3f s, fit %.3f s (%d iterations): %.3f s end to end"
              % (type(geom).__name__, n, N, t1 - t0, t2 - t1, geom.inc, geom.PA, t3 - t2, t4 - t3, FF.iteration_diagnostics["num_iterations"], t4 - t0), flush=True)

"""predict_visibilities at 1e7 points: bucket-table path vs the direct J0 kernel (development tool)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import DiscreteHankelTransform, FixedGeometry, VisibilityMapping
from frank_amd.constants import rad_to_arcsec
from frank_amd.mock import MOCK_GEOMETRY
vm = VisibilityMapping(DiscreteHankelTransform(2.0 / rad_to_arcsec, 300), FixedGeometry(**MOCK_GEOMETRY), verbose=False)
rng = np.random.default_rng(0)
q = np.exp(rng.uniform(np.log(1e4), np.log(2e6), 10 ** 7))
I = np.exp(-0.5 * (vm.r / 0.3) ** 2) * 1e10
for rep in range(3):
    t = time.perf_counter(); V = vm.predict_visibilities(I, q); dt = time.perf_counter() - t
    print("%s: predict 1e7 points, N=300: %.1f ms incl. 160 MB of PCIe" % (os.environ.get("FRANK_AMD_K1", "tables"), dt * 1e3))

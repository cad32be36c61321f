import sys, os, ctypes, numpy as np, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from frank_amd import _lib as L
from frank_amd.mock import mock_disc_visibilities
f = bench.Fitter(L, 300, 0)
u, v, V, w = mock_disc_visibilities(10_000_000, seed=0, noise_seed=50)
f.upload(u, v, V, w)
for i in range(3):
    f.bin(); f.sync(); print(os.environ.get("FRANK_AMD_LIB","default").split("_")[-1], "bin_gram kernel ms", f.kernel_ms())

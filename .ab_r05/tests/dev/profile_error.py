import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from frank_amd import FrankFitter, FixedGeometry
from frank_amd.constants import rad_to_arcsec
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities
from oracle import oracle as fo
tag = os.environ.get("FRANK_AMD_K1", "moments")
N, n = 60, 20000
u, v, V, w = mock_disc_visibilities(n, seed=21, noise_seed=22)
FF = FrankFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), alpha=1.3, weights_smooth=1e-2, verbose=False)
pre = FF.preprocess_visibilities(u, v, V, w)
sol = FF.fit_preprocessed(pre)
g = MOCK_GEOMETRY
m = fo.map_visibilities(N, 2.0 / rad_to_arcsec, (g["inc"], g["PA"], g["dRA"], g["dDec"]), u, v, V, w)
ref = fo.frank_fit_normal(N, 2.0 / rad_to_arcsec, m["M"], m["j"], alpha=1.3, wsmooth=1e-2)
print(tag, "smoke case: M %.2e j %.2e profile %.2e" % (np.abs(pre["M"]-m["M"]).max()/np.abs(m["M"]).max(), np.abs(pre["j"]-m["j"]).max()/np.abs(m["j"]).max(), np.abs(sol.I-ref["mu"]).max()/np.abs(ref["mu"]).max()))
# same M into both fits: is it the fit or the mapping?
FF._M, FF._j = m["M"], m["j"]
s2 = FF._fit()
print(tag, "  oracle's M, j through the device fit: profile %.2e" % (np.abs(s2.I-ref["mu"]).max()/np.abs(ref["mu"]).max()))
for name, NN in (("fit_N100_1e5.npz", 100), ("fit_N300_1e6.npz", 300)):
    gg = np.load("tests/golden/" + name)
    u, v, V, w = mock_disc_visibilities(int(gg["n"]), seed=int(gg["seed"]), noise_seed=int(gg["noise_seed"]))
    F = FrankFitter(2.0, NN, FixedGeometry(**MOCK_GEOMETRY), verbose=False)
    s = F.fit(u, v, V, w)
    print(tag, name, "profile vs the reference %.2e" % (np.abs(s.I - gg["I"]).max() / np.abs(gg["I"]).max()))

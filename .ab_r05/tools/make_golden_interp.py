#!/usr/bin/env python3
"""Golden fixture of the Fourier-Bessel interpolation (hankel.py:206-263, statistical_models.py:435-481,
radial_fitters.py:146-176): imports the REFERENCE (build container only) and records inputs and outputs.
    python3 tools/make_golden_interp.py
"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, "/root/reference")
import scipy  # noqa: E402
import frank  # noqa: E402
from frank.constants import rad_to_arcsec  # noqa: E402
from frank.geometry import FixedGeometry  # noqa: E402
from frank.hankel import DiscreteHankelTransform  # noqa: E402
from frank.statistical_models import VisibilityMapping  # noqa: E402

out = {}
for N, Rmax in ((100, 5.0), (300, 2.0 / rad_to_arcsec)):
    d = DiscreteHankelTransform(Rmax, N)
    rng = np.random.default_rng(N)
    rpts = np.concatenate([[0.0], np.sort(rng.uniform(0, 1.2 * Rmax, 40)), [0.5 * (d.r[3] + d.r[4])]])
    qpts = np.concatenate([[0.0], np.sort(rng.uniform(0, 1.2 * d.Qmax, 40))])
    f = np.exp(-0.5 * (d.r / (0.2 * Rmax)) ** 2) * (1 + 0.3 * np.cos(9 * d.r / Rmax))
    g = np.exp(-0.5 * (d.q / (0.2 * d.Qmax)) ** 2)
    out["N%d_rpts" % N], out["N%d_qpts" % N], out["N%d_f" % N], out["N%d_g" % N] = rpts, qpts, f, g
    out["N%d_Yreal" % N] = d.interpolation_coefficients(rpts, "Real")
    out["N%d_Yfourier" % N] = d.interpolation_coefficients(qpts, "Fourier")
    out["N%d_freal" % N] = d.interpolate(f, rpts, "Real")
    out["N%d_gfourier" % N] = d.interpolate(g, qpts, "Fourier")
# the gaussian of the reference's test_hankel_gauss (tests.py:37-82), interpolated between its collocation points
d = DiscreteHankelTransform(5.0, 100)
r = np.linspace(0, 5.0, 25)
out["gauss_r"] = r
out["gauss_interp"] = d.interpolate(np.exp(-0.5 * d.r ** 2), r, "Real")
# VisibilityMapping.interpolate: arcsec in, any shape, chunked
d = DiscreteHankelTransform(2.0 / rad_to_arcsec, 50)
vm = VisibilityMapping(d, FixedGeometry(30., 40., 0., 0.), block_size=700, verbose=False)
R = np.linspace(0.0, 1.9, 24).reshape(4, 6)
I = np.exp(-0.5 * ((d.r * rad_to_arcsec - 0.6) / 0.2) ** 2)
out["vm_R"], out["vm_I"], out["vm_out"] = R, I, vm.interpolate(I, R, space="Real")
path = os.path.join(ROOT, "tests", "golden", "interpolate.npz")
np.savez_compressed(path, meta_reference_version=frank.__version__, meta_numpy=np.__version__, meta_scipy=scipy.__version__, **out)
print("wrote", path, os.path.getsize(path) // 1024, "KB")

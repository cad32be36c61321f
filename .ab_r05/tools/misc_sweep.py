"""Three more sweeps over sizes (independent implementations against each other; exit status 1 on a mismatch):
 (a) a single-precision table (fh_vis_upload_f32) against the double table holding the widened values: M, j, H0 BIT-identical,
     over basis sizes x row counts;
 (b) DiscreteHankelTransform.coefficients(q) and VisibilityMapping.predict_visibilities (device Bessel / bucket tables)
     against scipy.special.j0 on the host, over basis sizes x numbers of baselines, q from 0 to beyond Qmax;
 (c) UVDataBinner (LDS histograms, global atomics above 3072 bins) against numpy.bincount over bin counts x row counts:
     counts exact, weighted means to 1e-10;
 (d) sol.predict(u, v) through the bucket tables against the direct evaluation over basis sizes x call sizes.
     python3 tools/misc_sweep.py"""
import os
import sys

import numpy as np
from scipy.special import j0

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import DiscreteHankelTransform, FixedGeometry, VisibilityMapping  # noqa: E402
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities  # noqa: E402
from frank_amd.utilities import UVDataBinner  # noqa: E402

bad = []
geom = FixedGeometry(**MOCK_GEOMETRY)
u, v, V, w = mock_disc_visibilities(300001, seed=41, noise_seed=42)
w = np.full(u.size, w) if np.ndim(w) == 0 else w
u4, v4, V4, w4 = u.astype(np.float32), v.astype(np.float32), V.astype(np.complex64), w.astype(np.float32)
for N in (3, 17, 100, 255, 300, 304, 511, 700):
    vm = VisibilityMapping(DiscreteHankelTransform(2.0 / 206264.80624709636, N), geom, check_qbounds=False)
    for n in (1, 63, 1000, 65537, 300001):
        a = vm.map_visibilities(u4[:n], v4[:n], V4[:n], w4[:n])
        b = vm.map_visibilities(u4[:n].astype(np.float64), v4[:n].astype(np.float64), V4[:n].astype(np.complex128), w4[:n].astype(np.float64))
        ok = np.array_equal(a["M"], b["M"]) and np.array_equal(a["j"], b["j"]) and a["null_likelihood"] == b["null_likelihood"]
        if not ok:
            bad.append(("f32", N, n))
        print("f32 table N=%3d n=%6d  %s" % (N, n, "identical" if ok else "DIFFERENT  <-- MISMATCH"), flush=True)

rng = np.random.default_rng(3)
for N in (2, 3, 16, 100, 300, 511, 640, 1000):
    D = DiscreteHankelTransform(2.0 / 206264.80624709636, N)
    vm = VisibilityMapping(D, geom, vis_model='opt_thin', check_qbounds=False)  # (scale 1: no cos(inc))
    I = np.exp(-0.5 * (D.r / (0.4 * D.Rmax)) ** 2) * (1.5 + np.sin(7 * D.r / D.Rmax))
    for nq in (1, 2, 63, 1000, 100003):
        q = np.concatenate([[0.0], rng.uniform(0, 1.3 * D.Qmax, nq - 1)]) if nq > 1 else np.array([0.37 * D.Qmax])
        H = D._scale_factor / (np.pi * D.Qmax ** 2) * j0(np.outer(q / D.Qmax, D._j_nk)) if nq <= 1000 else None
        Vh = (H @ I) if H is not None else None
        Vd = vm.predict_visibilities(I, q)
        if H is not None:
            Hd = D.coefficients(q)
            eH = np.abs(Hd - H).max() / np.abs(H).max()
            eV = np.abs(Vd - Vh).max() / np.abs(Vh).max()
        else:  # the big one: against coefficients() in slices (different kernel from predict)
            eH = 0.0
            Vc = np.concatenate([D.coefficients(q[s:s + 5000]) @ I for s in range(0, nq, 5000)])
            eV = np.abs(Vd - Vc).max() / np.abs(Vc).max()
        ok = eH < 1e-12 and eV < 1e-11
        if not ok:
            bad.append(("predict", N, nq))
        print("predict N=%4d nq=%6d  H %.1e  V %.1e %s" % (N, nq, eH, eV, "" if ok else "  <-- MISMATCH"), flush=True)

for n in (1, 2, 1000, 1000003):
    q = np.exp(rng.uniform(np.log(1e4), np.log(2e6), n))
    Vq = rng.normal(size=n) + 1j * rng.normal(size=n)
    wq = rng.uniform(0.5, 2.0, n)
    for nb in (1, 2, 7, 100, 3071, 3072, 3073, 5000, 20000, 200001):
        bw = q.max() / nb * (1 + 1e-9)
        b = UVDataBinner(q, Vq, wq, bw)
        idx = np.floor(q / bw).astype(np.int64)
        nbin = len(b)
        cnt = np.bincount(idx, minlength=nbin)[:nbin]
        sw = np.bincount(idx, weights=wq, minlength=nbin)[:nbin]
        sV = (np.bincount(idx, weights=wq * Vq.real, minlength=nbin) + 1j * np.bincount(idx, weights=wq * Vq.imag, minlength=nbin))[:nbin]
        got = np.ma.filled(b.bin_counts, 0)
        okc = np.array_equal(got, cnt)
        m = cnt > 0
        eV = np.abs(np.ma.filled(b.V, 0)[m] - sV[m] / sw[m]).max() if okc else np.inf
        ew = np.abs(np.ma.filled(b.weights, 0)[m] / sw[m] - 1).max() if okc else np.inf
        ok = okc and eV < 1e-10 and ew < 1e-12
        if not ok:
            bad.append(("uvbin", n, nb))
        print("uvbin n=%7d bins=%6d (len %6d)  counts %s  V %.1e  w %.1e %s" % (n, nb, nbin, "exact" if okc else "DIFFER", eV, ew, "" if ok else "  <-- MISMATCH"), flush=True)
# (d) sky-plane predict (fh_predict_sky): large calls go through the bucket tables, FRANK_AMD_RESIDUAL_DIRECT=1 keeps the N Bessel
#     evaluations per row; call sizes either side of the switch, basis sizes either side of the fused kernels' limit
from frank_amd.radial_fitters import FrankGaussianFit  # noqa: E402


class _Profile(FrankGaussianFit):
    def __init__(self, vm, I, geometry):
        FrankGaussianFit.__init__(self, vm, None, {}, geometry=geometry)
        self._I = I
    MAP = property(lambda self: self._I)


uu, vv, _, _ = mock_disc_visibilities(300001, seed=43, noise_seed=44)
for N in (3, 20, 300, 511, 640, 1000):
    D = DiscreteHankelTransform(2.0 / 206264.80624709636, N)
    vm = VisibilityMapping(D, geom, check_qbounds=False)
    I = np.exp(-0.5 * (D.r / (0.4 * D.Rmax)) ** 2) * (1.5 + np.sin(7 * D.r / D.Rmax))
    sol = _Profile(vm, I, geom)
    for nq in (1, 65535, 65536, 300001):
        P = sol.predict(uu[:nq], vv[:nq])
        os.environ["FRANK_AMD_RESIDUAL_DIRECT"] = "1"
        Pd = sol.predict(uu[:nq], vv[:nq])
        del os.environ["FRANK_AMD_RESIDUAL_DIRECT"]
        e = np.abs(P - Pd).max() / np.abs(Pd).max()
        ok = e < 1e-11
        if not ok:
            bad.append(("predict_sky", N, nq))
        print("predict_sky N=%4d n=%6d  tables vs direct %.1e %s" % (N, nq, e, "" if ok else "  <-- MISMATCH"), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)

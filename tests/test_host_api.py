"""CPU-only checks of the product's host side: the C-ABI library loads, exports every symbol that
include/frank_hip.h declares, its host DHT set-up matches the reference fixtures, and the device
entry points fail loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import pickle
import re

import numpy as np
import pytest

from conftest import ROOT, rel_to_max
from frank_amd.constants import rad_to_arcsec

RMAX = 2.0 / rad_to_arcsec


def test_library_exports_every_declared_symbol():
    from frank_amd import _lib
    header = open(os.path.join(ROOT, "include", "frank_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(fh_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), "libfrank_hip.so does not export %s" % name
    assert declared == set(_lib.SIGNATURES), "frank_amd/_lib.py and include/frank_hip.h disagree"


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under frank_amd/ may reference it."""
    pkg = os.path.join(ROOT, "frank_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "frank_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


@pytest.mark.parametrize("N", [5, 20, 100, 300])
def test_host_dht_matches_reference(golden, N):
    from frank_amd import DiscreteHankelTransform
    g = golden("dht_N%d.npz" % N)
    d = DiscreteHankelTransform(RMAX, N)
    assert d.size == N and d.order == 0
    zeros = np.append(d._j_nk, d._j_nN)
    # the collocation grid is bit-identical with the reference's (hankel.py:72-78): same zeros (SciPy's values are
    # tabulated, tools/gen_j0_zeros_table.py), same fp64 expressions in the same order
    assert np.array_equal(zeros, g["zeros"])
    assert np.array_equal(d.r, g["r"]) and np.array_equal(d.q, g["q"])
    assert d.Qmax == float(g["Qmax"])
    np.testing.assert_allclose(d._scale_factor, g["scale_factor"], rtol=2e-13)
    assert np.abs(d._Ykm - g["Ykm"]).max() <= 2e-14 * np.abs(g["Ykm"]).max()
    assert rel_to_max(d.coefficients(), g["Y"]) < 1e-13
    assert rel_to_max(d.transform(np.ones(N)), g["transform_ones"]) < 1e-12
    with pytest.raises(AttributeError):
        d.coefficients(direction="sideways")             # hankel.py:194
    with pytest.raises(AttributeError):
        d.transform(np.ones(N), direction="sideways")    # hankel.py:159
    d2 = pickle.loads(pickle.dumps(d))
    assert np.array_equal(d2.q, d.q) and np.array_equal(d2._Ykm, d._Ykm)


def test_collocation_points_reference_literals():
    """frank/tests.py:704-717"""
    from frank_amd import DiscreteHankelTransform
    r, q = DiscreteHankelTransform.get_collocation_points(RMAX, 10)
    np.testing.assert_allclose(r * rad_to_arcsec, [0.14239924, 0.32686567, 0.51242148, 0.69822343, 0.88411873,
                                                   1.07005922, 1.25602496, 1.44200623, 1.62799772, 1.8139963],
                               rtol=2e-5, atol=1e-8)
    np.testing.assert_allclose(q, [39472.88305737, 90606.73736504, 142042.56471889, 193546.62066389,
                                   245076.55732463, 296619.01772663, 348168.47711355, 399722.24089812,
                                   451278.83939289, 502837.4032234], rtol=2e-5, atol=1e-8)


def test_smoothing_matrix_host_mirror(golden):
    from frank_amd import DiscreteHankelTransform
    from frank_amd.filter import spectral_smoothing_matrix
    g = golden("smoothing_T.npz")
    for N in (20, 100):
        T = spectral_smoothing_matrix(DiscreteHankelTransform(RMAX, N), float(g["w_N%d" % N]))
        assert rel_to_max(T, g["T_N%d" % N]) < 1e-12


def test_constructor_errors():
    from frank_amd import FixedGeometry, FrankFitter, VisibilityMapping, DiscreteHankelTransform
    geom = FixedGeometry(30.0, 40.0)
    with pytest.raises(ValueError):
        FrankFitter(2.0, 20, geom, method="Cauchy")                  # radial_fitters.py:697-699
    with pytest.raises(ValueError):
        FrankFitter(2.0, 20, geom, convergence_failure="explode")    # radial_fitters.py:728-730
    with pytest.raises(ValueError):
        VisibilityMapping(DiscreteHankelTransform(RMAX, 10), geom, vis_model="opaque")  # statistical_models.py:71-73
    FF = FrankFitter(2.0, 20, geom, verbose=False)
    assert FF._info == {'Rmax': pytest.approx(2.0), 'N': 20, 'alpha': 1.05, 'wsmooth': 1e-4, 'p0': 1e-15,
                        'method': 'Normal'}
    assert FF.Rmax == pytest.approx(2.0) and FF.size == 20 and FF.geometry is geom


def test_device_entry_points_fail_loudly_without_a_gpu():
    from frank_amd import _lib, FixedGeometry, FrankFitter
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    FF = FrankFitter(2.0, 20, FixedGeometry(30.0, 40.0), verbose=False)
    x = np.full(8, 1e5)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FF.fit(x, x, np.ones(8) + 0j, np.ones(8))
    x4 = x.astype(np.float32)  # single-precision tables take the fp32 upload: the same loud failure
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FF.fit(x4, x4, (np.ones(8) + 0j).astype(np.complex64), np.ones(8, dtype=np.float32))
    # the other entry families: LogNormal, uv-binner, bootstrap
    from frank_amd.utilities import UVDataBinner
    from frank_amd.bootstrap import bootstrap_fits
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        UVDataBinner(x, np.ones(8) + 0j, np.ones(8), 1e4)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        bootstrap_fits(FF, x, x, np.ones(8) + 0j, np.ones(8), 1)
    FL = FrankFitter(2.0, 20, FixedGeometry(30.0, 40.0), method="LogNormal", verbose=False)
    assert FL._info["p0"] == 1e-35 and FL.fit_method() == "FrankFitter: LogNormal method"
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        FL.fit(x, x, np.ones(8) + 0j, np.ones(8))


def test_constants_are_the_reference_doubles():
    """frank/constants.py:23-25 evaluated as NumPy would: the same IEEE doubles."""
    import frank_amd.constants as c
    assert c.rad_to_arcsec == 3600 * 180 / np.pi
    assert c.sterad_to_arcsec == (3600 * 180 / np.pi) ** 2
    assert c.deg_to_rad == np.pi / 180


def test_bin_gram_tile_tables_cover_the_triangle_once():
    """The tile-ownership tables of bin_gram_kernel<19> (kTiles19P0 / kTiles19P1) are a partition of the 190 upper-triangle
    tiles: part 0 the block rows 0-6 (tiles 0..111), part 1 the rest; at most 10 tiles per wave; equal (+-1) tile counts
    per SIMD (waves W, W+4, W+8)."""
    src = open(os.path.join(ROOT, "frank_amd", "csrc", "bin_gram.hip")).read()
    for name, lo, hi in (("kTiles19P0", 0, 112), ("kTiles19P1", 112, 190)):
        body = re.search(r"constexpr short %s\[12\]\[10\] = \{(.*?)\n\};" % name, src, flags=re.S).group(1)
        rows = re.findall(r"\{([^{}]*)\}", body)
        assert len(rows) == 12
        waves = [[int(x) for x in r.split(",") if int(x) >= 0] for r in rows]
        assert all(1 <= len(w) <= 10 for w in waves)
        assert sorted(t for w in waves for t in w) == list(range(lo, hi))
        per_simd = [sum(len(waves[w]) for w in (s, s + 4, s + 8)) for s in range(4)]
        assert max(per_simd) - min(per_simd) <= 1


def test_split_grid_and_device_argument():
    """Host logic of the multi-GPU paths: the sweep grid is dealt in contiguous near-equal slices, and the
    `device` of a transform survives pickling (result objects hold no native handles)."""
    import pickle
    from frank_amd import DiscreteHankelTransform
    from frank_amd.sweep import split_grid
    assert split_grid(512, 8) == [(64 * d, 64) for d in range(8)]
    assert split_grid(7, 3) == [(0, 3), (3, 2), (5, 2)]
    assert split_grid(2, 4) == [(0, 1), (1, 1), (2, 0), (2, 0)]
    d = DiscreteHankelTransform(1e-5, 12, device=5)
    assert d.device == 5
    d2 = pickle.loads(pickle.dumps(d))
    assert d2.device == 5 and np.array_equal(d2.q, d.q)
    os.environ["FRANK_AMD_DEVICE"] = "2"
    try:
        assert DiscreteHankelTransform(1e-5, 12).device == 2
    finally:
        del os.environ["FRANK_AMD_DEVICE"]


@pytest.mark.parametrize("N", [20, 300])
def test_bucket_tables_reproduce_j0(N):
    """The Taylor tables bin_gram multiplies on the matrix pipe (fh_dht_bucket_tables, host long double) against
    scipy.special.j0 and mpmath at random points of random buckets: |error| <= 2.5e-16 (Cephes itself is 4e-16 .. 1.3e-15
    off the true J0 in this range), for the first bucket (expansion point next to the singular point of Bessel's
    equation), the buckets of the bench workload and the last bucket of the q range."""
    import mpmath
    from scipy.special import j0
    from frank_amd import _lib
    d = ctypes.c_void_p()
    _lib.check(_lib.lib.fh_dht_create(RMAX, N, 0, ctypes.byref(d)))
    zeros = np.empty(N + 1)
    _lib.check(_lib.lib.fh_dht_get(d, None, None, _lib.ptr(zeros), None, None, None, None))
    delta = ctypes.c_double()
    _lib.check(_lib.lib.fh_dht_bucket_tables(d, 0, 0, None, ctypes.byref(delta)))
    assert delta.value == 0.5 / zeros[N - 1]
    nb = int(1.0 / delta.value) + 1
    rng = np.random.default_rng(7)
    worst_mp = 0.0
    for b in [0, 1, 2, nb // 7, nb // 2, nb - 1]:
        tab = np.empty((12, N))
        _lib.check(_lib.lib.fh_dht_bucket_tables(d, b, b + 1, _lib.ptr(tab), None))
        s0 = (b + 0.5) * delta.value
        tau = rng.uniform(-1, 1, 64)
        tau[:2] = (-1.0, 1.0)
        P = tau[:, None] ** np.arange(12)[None, :]
        X = np.zeros((64, N))
        for n in range(11, -1, -1):  # small terms first, as the kernel accumulates
            X += P[:, n:n + 1] * tab[n][None, :]
        s = s0 + tau * (delta.value / 2)
        ref = j0(np.outer(s, zeros[:N]))
        # scipy's own error (4e-16 .. 1.3e-15) plus the rounding of its argument fl(s * j_k), up to 1e-13 * |J1| at x ~ 1e3
        assert np.abs(X - ref).max() < 1e-14
        mpmath.mp.dps = 30
        for i, k in ((0, 0), (1, N - 1), (5, N // 3), (9, N // 2), (17, 1)):
            x = (mpmath.mpf(s0) + mpmath.mpf(float(tau[i])) * mpmath.mpf(delta.value / 2)) * mpmath.mpf(float(zeros[k]))
            worst_mp = max(worst_mp, abs(float(mpmath.besselj(0, x) - mpmath.mpf(float(X[i, k])))))
    assert worst_mp < 2.5e-16
    _lib.lib.fh_dht_destroy(d)


def test_geometry_fit_and_debris_classes_host_side():
    """Signatures of frank/geometry.py:430, :643-644 and frank/debris_fitters.py:57-58, :142-146 (plus this package's
    device=/arithmetic= keywords at the end); the range fold of geometry.py:33-39; and the call form that needs no fit."""
    import inspect
    from frank_amd.geometry import FitGeometryGaussian, FitGeometryFourierBessel, _fix_inc_and_PA_ranges
    from frank_amd import debris_fitters as dfit

    def names(f):
        return list(inspect.signature(f).parameters)[1:]
    assert names(FitGeometryGaussian.__init__)[:3] == ["inc_pa", "phase_centre", "guess"]
    assert names(FitGeometryFourierBessel.__init__)[:6] == ["Rmax", "N", "inc_pa", "phase_centre", "guess", "verbose"]
    assert names(dfit.FourierBesselDebrisFitter.__init__)[:8] == ["Rmax", "N", "geometry", "scale_height", "nu", "block_data",
                                                                  "block_size", "verbose"]
    assert names(dfit.FrankDebrisFitter.__init__)[:18] == [
        "Rmax", "N", "geometry", "scale_height", "nu", "block_data", "block_size", "alpha", "p_0", "weights_smooth", "tol",
        "method", "I_scale", "max_iter", "check_qbounds", "store_iteration_diagnostics", "verbose", "convergence_failure"]
    assert _fix_inc_and_PA_ranges(190.0, 200.0) == (10.0, 20.0)
    assert _fix_inc_and_PA_ranges(100.0, -10.0) == (80.0, 170.0)
    assert _fix_inc_and_PA_ranges(-20.0, 180.0) == (20.0, 0.0)
    x = np.zeros(4)
    for g in (FitGeometryGaussian(inc_pa=(30.0, 40.0), phase_centre=(0.1, -0.2)),
              FitGeometryFourierBessel(2.0, 20, inc_pa=(30.0, 40.0), phase_centre=(0.1, -0.2))):
        g.fit(x, x, x, x + 1)  # both pairs given: nothing is fitted, nothing touches the device
        assert (g.inc, g.PA, g.dRA, g.dDec) == (30.0, 40.0, 0.1, -0.2)
        c = g.clone()
        assert (c.inc, c.PA, c.dRA, c.dDec) == (30.0, 40.0, 0.1, -0.2)
    for bad in (lambda: FitGeometryGaussian(optimizer="minuit"), lambda: FitGeometryFourierBessel(2.0, 20, optimizer="minuit")):
        with pytest.raises(ValueError):
            bad()
    # the default starting point and how given values overwrite it (geometry.py:437-447, :653-660)
    assert FitGeometryGaussian()._guess == [10.0, 10.0, 0.0, 0.0, 1.0, 1.0]
    assert FitGeometryGaussian(inc_pa=(5.0, 6.0), guess=[1.0, 2.0, 3.0, 4.0])._guess == [5.0, 6.0, 3.0, 4.0, 1.0, 1.0]
    assert FitGeometryFourierBessel(2.0, 20, phase_centre=(7.0, 8.0))._guess == [10.0, 10.0, 7.0, 8.0]


def test_levenberg_marquardt_on_normal_equations_follows_minpack():
    """frank_amd/_levmar.py against scipy.optimize.least_squares(method='lm') (MINPACK lmdif) on dense problems where the
    Jacobian can be formed on the host: same minimiser to 1e-9, same number of residual evaluations on the well-conditioned
    one; parameters of very different sizes; a zero residual; a start at the minimum."""
    from scipy.optimize import least_squares
    from frank_amd._levmar import forward_steps, levenberg_marquardt

    def run(r, x0):
        state = {}

        def trial(x):
            state["t"] = r(x)
            return float(state["t"] @ state["t"])

        def accept():
            state["b"] = state["t"]

        def normal(x):
            h = forward_steps(x)
            J = np.stack([(r(x + h[k] * np.eye(x.size)[k]) - state["b"]) / h[k] for k in range(x.size)], axis=1)
            return J.T @ J, J.T @ state["b"], x.size
        return levenberg_marquardt(trial, accept, normal, x0)

    rng = np.random.default_rng(0)
    t = np.linspace(0, 4, 400)

    def model(x):
        return x[0] * np.exp(-x[1] * t) + x[2] * np.sin(x[3] * t)
    y = model([2.0, 1.3, 0.5, 3.0]) + 0.01 * rng.normal(size=t.size)
    x, info, nfev = run(lambda x: model(x) - y, [1.0, 1.0, 1.0, 2.5])
    ref = least_squares(lambda x: model(x) - y, [1.0, 1.0, 1.0, 2.5], method="lm")
    assert info in (1, 2, 3) and np.abs(x - ref.x).max() < 1e-9 and nfev == ref.nfev
    # parameters 1e4 apart in size, as an inclination in degrees and a phase centre in arcsec are

    def model2(x):
        return np.exp(-0.5 * (t - x[0] / 10.0) ** 2) * np.cos(40.0 * x[1] * t) * x[2]
    y2 = model2([20.0, 2e-3, 1.5])
    x, info, _ = run(lambda x: model2(x) - y2 + 1e-3 * np.sin(7 * t), [19.0, 1.8e-3, 1.2])
    ref = least_squares(lambda x: model2(x) - y2 + 1e-3 * np.sin(7 * t), [19.0, 1.8e-3, 1.2], method="lm")
    assert ref.status > 0 and info in (1, 2, 3, 4) and np.abs((x - ref.x) / ref.x).max() < 1e-6
    # ... and from a start where MINPACK itself runs out of evaluations in a valley: the same count, the same point
    x, info, nfev = run(lambda x: model2(x) - y2 + 1e-3 * np.sin(7 * t), [15.0, 1e-3, 1.0])
    ref = least_squares(lambda x: model2(x) - y2 + 1e-3 * np.sin(7 * t), [15.0, 1e-3, 1.0], method="lm")
    assert ref.status == 0 and info == 5 and nfev == ref.nfev and np.abs((x - ref.x) / ref.x).max() < 1e-3
    # an exact fit (zero residual at the minimum) and a start at the minimum
    x, info, _ = run(lambda x: model(x) - model([2.0, 1.3, 0.5, 3.0]), [1.8, 1.2, 0.6, 2.9])
    assert info in (1, 2, 3, 4) and np.abs(x - [2.0, 1.3, 0.5, 3.0]).max() < 1e-7
    x, info, _ = run(lambda x: model(x) - y, ref.x if ref.x.size == 4 else x)
    assert info in (1, 2, 3, 4)
    # a parameter the residual does not depend on (a zero Jacobian column, singular J^T J): it stays where it was, as in MINPACK
    y3 = 1.7 * np.exp(-0.8 * t)
    x, info, nfev = run(lambda x: x[0] * np.exp(-x[1] * t) - y3 + 0 * x[2], [1.0, 1.0, 5.0])
    ref = least_squares(lambda x: x[0] * np.exp(-x[1] * t) - y3 + 0 * x[2], [1.0, 1.0, 5.0], method="lm")
    assert info in (1, 2, 3, 4) and np.abs(x - ref.x).max() < 1e-9 and x[2] == 5.0 and nfev == ref.nfev
    assert np.array_equal(forward_steps(np.array([0.0, -2.0])), np.sqrt(np.finfo(float).eps) * np.array([1.0, 2.0]))


def test_header_is_plain_c():
    """include/frank_hip.h is the boundary: it must compile as C99 on its own (no C++, no HIP, no torch types)."""
    import subprocess
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".c", delete=False) as fh:
        fh.write('#include "frank_hip.h"\nint main(void) { return FH_RESIDUAL_SLOTS > 0 ? 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                        "-I", os.path.join(ROOT, "include"), fh.name], capture_output=True, text=True)
    os.unlink(fh.name)
    assert r.returncode == 0, r.stderr


def test_mock_data_helper_signatures_and_errors():
    """utilities.py:634, :923, :962-963, :1041, :1077-1078: argument names and defaults; the argument errors are raised
    before anything touches the device."""
    import inspect
    from frank_amd import utilities as ut
    from frank_amd.geometry import FixedGeometry

    def sig(f):
        return [(p.name, p.default) for p in inspect.signature(f).parameters.values()]
    E = inspect.Parameter.empty
    assert sig(ut.draw_bootstrap_sample) == [("u", E), ("v", E), ("vis", E), ("weights", E)]
    assert sig(ut.add_vis_noise) == [("vis", E), ("weights", E), ("seed", None)]
    assert sig(ut.get_collocation_points) == [("Rmax", 2.0), ("N", 500), ("direction", "forward")]
    assert sig(ut.generic_dht) == [("x", E), ("f", E), ("Rmax", 2.0), ("N", 500), ("direction", "forward"), ("grid", None), ("inc", 0.0)]
    assert sig(ut.make_mock_data) == [("r", E), ("I", E), ("Rmax", E), ("u", E), ("v", E), ("projection", None), ("geometry", None),
                                      ("N", 500), ("add_noise", False), ("weights", None), ("seed", None)]
    x = np.linspace(0.0, 1.0, 8)
    for bad in (dict(projection="sideways"), dict(projection="deproject"), dict(geometry=FixedGeometry(10.0, 20.0))):
        with pytest.raises(AttributeError):
            ut.make_mock_data(x, x, 2.0, x, x, **bad)
    with pytest.raises(AttributeError):
        ut.generic_dht(x, x, direction="sideways")
    with pytest.raises(AttributeError):
        ut.get_collocation_points(direction="sideways")
    a = ut.add_vis_noise(np.ones(5), 4.0 * np.ones(5), seed=1)
    np.random.seed(1)
    assert np.array_equal(a, 1.0 + 0.5 * np.random.standard_normal((1, 5))[0])


def test_unit_and_cut_helpers():
    """utilities.py:31-177, 403-512 (host arithmetic; checked identical to the reference's functions on random inputs in the
    build container): known answers, round trips, the inclusive cut, the argument errors."""
    from frank_amd import utilities as ut
    from frank_amd.constants import sterad_to_arcsec
    from frank_amd.geometry import FixedGeometry
    assert ut.arcsec_baseline(1.0) == rad_to_arcsec and abs(ut.arcsec_baseline(ut.arcsec_baseline(0.37)) / 0.37 - 1) < 1e-15
    assert ut.radius_convert(0.5, 140.0) == 70.0 and ut.radius_convert(70.0, 140.0, 'au_arcsec') == 0.5
    with pytest.raises(AttributeError):
        ut.radius_convert(1.0, 1.0, 'pc_au')
    x = np.array([1.0, 2.5])
    beam = np.pi * 0.12 * 0.08 / (4 * np.log(2))
    assert np.array_equal(ut.jy_convert(x, 'beam_arcsec2', 0.12, 0.08), x / beam)
    assert np.array_equal(ut.jy_convert(x, 'arcsec2_sterad'), x * sterad_to_arcsec)
    for a, b in (('beam_sterad', 'sterad_beam'), ('beam_arcsec2', 'arcsec2_beam'), ('arcsec2_sterad', 'sterad_arcsec2')):
        np.testing.assert_allclose(ut.jy_convert(ut.jy_convert(x, a, 0.12, 0.08), b, 0.12, 0.08), x, rtol=1e-15)
    with pytest.raises(ValueError):
        ut.jy_convert(x, 'beam_sterad')
    with pytest.raises(AttributeError):
        ut.jy_convert(x, 'beam_parsec', 1.0, 1.0)
    u, v = np.array([3e4, 6e4, 0.0, 2e5]), np.array([4e4, 8e4, 1.5e5, 0.0])
    un, vn = ut.normalize_uv(u, v, 2.0)
    assert np.array_equal(un, u / 2) and np.array_equal(vn, v / 2)
    assert np.array_equal(ut.normalize_uv(u, v, [1.0, 2.0, 4.0, 8.0])[0], u / np.array([1.0, 2.0, 4.0, 8.0]))
    with pytest.raises(ValueError):
        ut.normalize_uv(u, v, [1.0, 2.0])
    V, w = np.arange(4) + 1j, np.ones(4)
    uc, vc, Vc, wc = ut.cut_data_by_baseline(u, v, V, w, [5e4, 1.5e5])   # baselines 5e4, 1e5, 1.5e5, 2e5: ends included
    assert np.array_equal(uc, u[:3]) and np.array_equal(Vc, V[:3]) and wc.size == 3
    g = FixedGeometry(60.0, 0.0)                                          # u compressed by cos(60 deg) = 1/2
    assert ut.cut_data_by_baseline(u, v, V, w, [0.0, 1.2e5], geometry=g)[0].size == 3

    class Fit(object):
        def __init__(self, method):
            self._info, self.covariance, self.I = {"method": method}, np.diag([0.04, 0.09]), np.array([2.0, 3.0])
    assert np.allclose(ut.get_fit_stat_uncer(Fit("Normal")), [0.2, 0.3])
    assert np.allclose(ut.get_fit_stat_uncer(Fit("LogNormal"), return_linear=False), [0.2, 0.3])
    assert np.allclose(ut.get_fit_stat_uncer(Fit("LogNormal")), np.sqrt((np.exp([0.04, 0.09]) - 1) * np.array([4.0, 9.0])))
    bad = Fit("Normal")
    bad._info = {}
    with pytest.raises(AttributeError):
        ut.get_fit_stat_uncer(bad)


def test_uvtable_files_round_trip(tmp_path, golden):
    """frank_amd.io (io.py:29-124): text, compressed text and npz UVTables; the header and column order of the text form
    (so that files written by the reference load, and the other way round); the extension errors."""
    import bz2
    import gzip
    from frank_amd import io as fio
    g = golden("realdata_multi_ring_N100.npz")
    u, v, w = g["u"][:200], g["v"][:200], g["w"][:200]
    V = g["Vre"][:200] + 1j * g["Vim"][:200]
    for name in ("t.txt", "t.dat", "t.npz"):
        path = str(tmp_path / name)
        fio.save_uvtable(path, u, v, V, w)
        got = fio.load_uvtable(path)
        for a, b in zip(got, (u, v, V, w)):
            assert np.array_equal(a, b), name  # (savetxt's %.18e round-trips a double)
    txt = open(str(tmp_path / "t.txt")).read()
    assert txt.splitlines()[0] == "# u [lambda]\tv [lambda]\tRe(V)  [Jy]\tIm(V) [Jy]\tWeight [Jy^-2]"
    assert len(txt.splitlines()[1].split()) == 5
    for opener, ext in ((gzip.open, ".gz"), (bz2.open, ".bz2")):
        with opener(str(tmp_path / ("t.txt" + ext)), "wt") as fh:
            fh.write(txt)
        assert np.array_equal(fio.load_uvtable(str(tmp_path / ("t.txt" + ext)))[2], V)
    np.savez(str(tmp_path / "real.npz"), u=u, v=v, V=V.real, weights=w)
    for bad in ("t.csv", "t.npz.gz", "real.npz"):
        with pytest.raises(ValueError):
            fio.load_uvtable(str(tmp_path / bad))
    with pytest.raises(ValueError):
        fio.save_uvtable(str(tmp_path / "t.csv"), u, v, V, w)
    with pytest.raises(ValueError):
        fio.save_fit(u, v, V, w, None, str(tmp_path / "x"), format="csv")


def test_no_transcendental_result_is_read_in_the_next_slot():
    """tile_chol.h ends its hand-written column block on v_rcp_f64 (+ s_nop 0): a VALU read of a transcendental's result needs one
    wait state and the compiler's hazard recogniser does not look into inline asm.  tools/check_trans_hazard.py compiles
    fit_loop.hip to assembly and scans every transcendental instruction for a reader in the slot directly behind it."""
    import subprocess
    import sys
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_trans_hazard.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 read in the next slot" in r.stdout


def test_no_accumulator_of_an_inline_matrix_instruction_is_read_too_early():
    """fit_loop.hip, rr_mfma4_*: the register-resident fit loop issues its tile products from inline asm with the accumulator tied in
    place (the builtin left the destination to the register allocator: whole-tile copies and scratch under 25 live tiles), and the
    hazard recogniser does not look into inline asm -- a copy or spill of the tile placed directly behind a block reads registers
    the matrix pipe has not written yet (seen: a Cholesky that failed at step 5 or 4 depending on the build).  The blocks end on 18
    wait states; tools/check_mfma_hazard.py compiles fit_loop_rr.hip to assembly and checks every block's surroundings."""
    import subprocess
    import sys
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_mfma_hazard.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 followed by a read" in r.stdout
    # ... and the check bites: without the wait states the same scan finds readers
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_mfma_hazard.py"), "-DRR_NO_BLOCK_NOPS"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 1, r.stdout + r.stderr


def test_shipped_library_reads_only_the_form_switches():
    """Round 6 hygiene: the tuning knobs, probes and kill-line forms of the earlier rounds (~25 FRANK_AMD_* variables, one of them a
    kernel whose results are garbage) are compiled out of the shipped library (capi_internal.h: FH_DEV_*; `make dev` builds them in).
    What is left selects among forms that give the same results and is listed in INTEGRATION.md."""
    import re
    from frank_amd import _lib
    with open(_lib.LIB_PATH, "rb") as fh:
        names = set(m.decode() for m in re.findall(rb"FRANK_AMD_[A-Z0-9_]+", fh.read()))
    allowed = {"FRANK_AMD_K1", "FRANK_AMD_K1_FUSED", "FRANK_AMD_K1_NO_HIST_CACHE", "FRANK_AMD_K1_SAFE_TRIG", "FRANK_AMD_NO_RANGE_CACHE",
               "FRANK_AMD_K2", "FRANK_AMD_K2_CLUSTER", "FRANK_AMD_K2_CLUSTER_BREAK", "FRANK_AMD_K2_RR", "FRANK_AMD_K2_DEFER",
               "FRANK_AMD_K2_LL", "FRANK_AMD_SWEEP_CAP", "FRANK_AMD_SWEEP_LEFT", "FRANK_AMD_SWEEP_STAGE2_CLUSTERS",
               "FRANK_AMD_SWEEP_NO_CLUSTERS", "FRANK_AMD_LN_CLUSTER", "FRANK_AMD_LN_CLUSTER_CHOL", "FRANK_AMD_LN_PIVOTED", "FRANK_AMD_LN_WIDE", "FRANK_AMD_RESIDUAL_DIRECT"}
    assert names <= allowed, sorted(names - allowed)
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for n in names:
        assert n in text or n.replace("FRANK_AMD", "") in text, n + " is not documented in INTEGRATION.md"
    # importing the package leaves the environment alone (the hardware-queue variable is settled at the first device entry)
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-c", "import os; os.environ.pop('GPU_MAX_HW_QUEUES', None); import frank_amd; "
                          "from frank_amd import _lib; import ctypes; print(ctypes.CDLL(None).getenv(b'GPU_MAX_HW_QUEUES'))"],
                         capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert out.returncode == 0, out.stderr[-800:]
    assert out.stdout.strip().splitlines()[-1] in ("0", "None"), out.stdout


"""The callers and data formats either side of the hot path, against the reference's results on the same inputs: the geometry
fits (geometry.py:404-763, tests/golden/geometry_fits_2e4.npz) and the debris fitter classes; the mock-data helpers
(utilities.py:923-1146, mockdata_helpers.npz); the multi-frequency mapping (statistical_models.py:175-237,
multifreq_N50_2e4.npz); sol.predict(u, v) as one device pass; io.save_fit / load_sol."""
import ctypes
import hashlib

import numpy as np
import pytest

from frank_amd.constants import deg_to_rad, rad_to_arcsec
from frank_amd.mock import mock_disc_visibilities

pytestmark = pytest.mark.gpu


def table(g):
    u, v, V, w = mock_disc_visibilities(int(g["n"]), seed=int(g["seed"]), noise_seed=int(g["noise_seed"]),
                                        weight=float(g["weight"]), qmax=float(g["qmax"]))
    sha = hashlib.sha256(b"".join(np.ascontiguousarray(a).tobytes() for a in (u, v, V, w))).hexdigest()
    assert sha == str(g["input_sha256"])
    return u, v, V, w


def geometry_of(f):
    return np.array([f.inc, f.PA, f.dRA, f.dDec])


def test_fourier_bessel_residual_function(golden):
    """FitGeometryFourierBessel._residual at a trial geometry: bin + solve + sqrt(w) (predict - V), every 8th entry and the
    sum of squares against the reference's (1e-9 of the largest residual: the prior-free N = 20 solve is well conditioned)."""
    from frank_amd import DiscreteHankelTransform
    from frank_amd.geometry import FitGeometryFourierBessel, _ResidentTable
    g = golden("geometry_fits_2e4.npz")
    u, v, V, w = table(g)
    f = FitGeometryFourierBessel(float(g["Rmax"]), int(g["N"]), optimizer="scipy")
    DHT = DiscreteHankelTransform(float(g["Rmax"]) / rad_to_arcsec, int(g["N"]))
    t = _ResidentTable(DHT.device, u, v, V, w)
    r = f._residual(tuple(g["trial"]), uvdata=(DHT, t))
    ref = g["resid_every8"]
    assert r.shape == (2 * u.size,)
    assert np.abs(r[::8] - ref).max() < 1e-9 * np.abs(ref).max()
    assert abs(np.sum(r * r) / float(g["resid_sumsq"]) - 1) < 1e-10
    # (that went through the bucket tables of the binning pass before it; the same with N Bessel evaluations per row)
    import os
    os.environ["FRANK_AMD_RESIDUAL_DIRECT"] = "1"
    try:
        rd = f._residual(tuple(g["trial"]), uvdata=(DHT, t))
    finally:
        del os.environ["FRANK_AMD_RESIDUAL_DIRECT"]
    assert 0 < np.abs(r - rd).max() < 1e-11 * np.abs(ref).max()
    # the same function through the classes a user would combine by hand (FourierBesselFitter.fit + sol.predict)
    from frank_amd import FixedGeometry, FourierBesselFitter
    geom = FixedGeometry(*g["trial"])
    sol = FourierBesselFitter(float(g["Rmax"]), int(g["N"]), geom, verbose=False).fit(u, v, V, w)
    e = np.sqrt(w) * (sol.predict(u, v) - V)
    assert np.abs(r - np.concatenate([e.real, e.imag])).max() < 1e-9 * np.abs(ref).max()
    # a slice of the table, a real-valued fp32 table and the sum of squares alone
    from frank_amd import _lib
    I, ss, out = np.ascontiguousarray(sol.I), ctypes.c_double(), np.empty(2 * 1000)
    gg = _lib.make_geometry(geom)
    _lib.check(_lib.lib.fh_vis_residuals(DHT.context(), ctypes.byref(gg), 0, t.handle, 5000, 1000, _lib.ptr(I), _lib.ptr(out),
                                         ctypes.byref(ss)))
    assert np.abs(out[:1000] - e.real[5000:6000]).max() < 1e-9 * np.abs(ref).max()
    assert np.abs(out[1000:] - e.imag[5000:6000]).max() < 1e-9 * np.abs(ref).max()
    _lib.check(_lib.lib.fh_vis_residuals(DHT.context(), ctypes.byref(gg), 0, t.handle, 5000, 1000, _lib.ptr(I), None,
                                         ctypes.byref(ss)))
    assert abs(ss.value / np.sum(out * out) - 1) < 1e-12
    assert _lib.lib.fh_vis_residuals(DHT.context(), ctypes.byref(gg), 0, t.handle, 19500, 1000, _lib.ptr(I), None, None) != 0
    t.close()


def test_gaussian_residuals_and_jacobian():
    """fh_gauss_residuals against the NumPy expressions of geometry.py:535-585 written out here, all four fit_* forms."""
    from frank_amd import _lib
    from frank_amd.geometry import _ResidentTable
    u, v, V, w = mock_disc_visibilities(5000, seed=3, noise_seed=4)
    w = w * np.random.default_rng(1).uniform(0.5, 2.0, u.size)
    t = _ResidentTable(0, u, v, V, w)
    x = np.array([0.6, 1.4, 0.03, -0.02, 0.8, 0.7])
    fac, sw = 2 * np.pi / rad_to_arcsec, np.sqrt(w)

    def wrap(z):
        return np.concatenate([z.real, z.imag])
    inc, PA, dRA, dDec, norm, scal = x
    phi = dRA * fac * u + dDec * fac * v
    Vp = V * (np.cos(phi) - 1j * np.sin(phi))
    up, vp = u * np.cos(PA) - v * np.sin(PA), u * np.sin(PA) + v * np.cos(PA)
    uv = up * up * np.cos(inc) ** 2 + vp * vp
    G = np.exp(-0.5 * uv / (scal * rad_to_arcsec) ** 2)
    fun_ref = wrap(sw * (norm * G - Vp))
    nn = norm / (scal * rad_to_arcsec) ** 2
    dVp = 1j * sw * Vp * fac
    cols = [wrap(nn * sw * G * up * up * np.cos(inc) * np.sin(inc) + 0j), wrap(nn * sw * G * up * vp * (np.cos(inc) ** 2 - 1) / 2 + 0j),
            wrap(dVp * u), wrap(dVp * v), wrap(sw * G + 0j), wrap(nn * sw * G * uv / scal + 0j)]
    for fit_ip in (1, 0):
        for fit_ph in (1, 0):
            fun, jac, ss = np.empty(2 * u.size), np.empty((2 * u.size, 6)), ctypes.c_double()
            _lib.check(_lib.lib.fh_gauss_residuals(t.handle, _lib.ptr(x), fit_ip, fit_ph, _lib.ptr(fun), _lib.ptr(jac),
                                                   ctypes.byref(ss)))
            assert np.abs(fun - fun_ref).max() < 1e-12 * np.abs(fun_ref).max()
            assert abs(ss.value / np.sum(fun_ref ** 2) - 1) < 1e-12
            for k in range(6):
                want = cols[k] if not ((k < 2 and not fit_ip) or (k in (2, 3) and not fit_ph)) else np.zeros(2 * u.size)
                assert np.abs(jac[:, k] - want).max() <= 1e-12 * max(np.abs(cols[k]).max(), 1e-300), (fit_ip, fit_ph, k)
    t.close()


@pytest.mark.parametrize("optimizer", ["device", "scipy"])
@pytest.mark.parametrize("tag,kw", [("free", {}), ("incpa", dict(inc_pa=(34.97, 85.76))), ("phase", dict(phase_centre=(1.9e-3, 2.5e-3)))])
def test_geometry_fits_against_the_reference(golden, tag, kw, optimizer):
    """Both fits in the reference's three call forms, from the same starting point: the fitted (inc, PA) within 1e-4 deg and
    (dRA, dDec) within 1e-7 arcsec of the reference's -- with the reference's optimiser on residual vectors copied out
    ('scipy': the same SciPy routine and tolerances; the residuals differ in the last bits) and with Levenberg-Marquardt on
    the normal equations reduced on the device ('device', the default: MINPACK's algorithm and tolerances)."""
    kw = dict(kw, optimizer=optimizer)
    from frank_amd.geometry import FitGeometryFourierBessel, FitGeometryGaussian
    g = golden("geometry_fits_2e4.npz")
    u, v, V, w = table(g)
    fg = FitGeometryGaussian(guess=list(g["guess"]), **kw)
    fg.fit(u, v, V, w)
    fb = FitGeometryFourierBessel(float(g["Rmax"]), int(g["N"]), guess=list(g["guess"]), **kw)
    fb.fit(u, v, V, w)
    for got, ref in ((geometry_of(fg), g["gauss_" + tag]), (geometry_of(fb), g["fb_" + tag])):
        assert np.abs(got[:2] - ref[:2]).max() < 1e-4, (got, ref)
        assert np.abs(got[2:] - ref[2:]).max() < 1e-7, (got, ref)
    # what the fit is for: it recovers the geometry the visibilities were made with
    truth = np.array([34.97, 85.76, 1.9e-3, 2.5e-3])
    assert np.abs(geometry_of(fb)[:2] - truth[:2]).max() < 0.1 and np.abs(geometry_of(fb)[2:] - truth[2:]).max() < 5e-4


def test_gaussian_fit_from_the_default_starting_point(golden):
    """From (10, 10, 0, 0) the reference's Gaussian fit of this table ends face-on (a local minimum; PA is then free):
    the same happens here."""
    from frank_amd.geometry import FitGeometryGaussian
    g = golden("geometry_fits_2e4.npz")
    u, v, V, w = table(g)
    f = FitGeometryGaussian()
    f.fit(u, v, V, w)
    ref = g["gauss_default_guess"]
    assert f.inc < 0.05 and ref[0] < 0.05
    assert abs(f.dRA - ref[2]) < 2e-5 and abs(f.dDec - ref[3]) < 2e-5


def test_fitter_with_a_geometry_fit_and_debris_classes():
    """A fitter given a geometry that still has to be fitted fits it first (radial_fitters.py:562); the debris classes are
    the base classes with scale_height set."""
    from frank_amd import FixedGeometry, FrankFitter, FrankDebrisFitter, FourierBesselDebrisFitter, FourierBesselFitter
    from frank_amd.geometry import FitGeometryGaussian
    u, v, V, w = mock_disc_visibilities(20000, seed=71, noise_seed=72, weight=1e6, qmax=1e6)
    FF = FrankFitter(2.0, 60, FitGeometryGaussian(guess=[30.0, 80.0, 0.0, 0.0]), verbose=False)
    sol = FF.fit(u, v, V, w)
    assert abs(sol.geometry.inc - 35.14) < 0.05 and abs(sol.geometry.PA - 84.91) < 0.05
    ref = FrankFitter(2.0, 60, FixedGeometry(sol.geometry.inc, sol.geometry.PA, sol.geometry.dRA, sol.geometry.dDec),
                      verbose=False).fit(u, v, V, w)
    assert np.array_equal(sol.I, ref.I)

    def H(R):
        return 0.03 * (R + 0.1)
    geom = FixedGeometry(34.97, 85.76, 1.9e-3, 2.5e-3)
    a = FrankDebrisFitter(2.0, 40, geom, H, verbose=False).fit(u, v, V, w)
    b = FrankFitter(2.0, 40, geom, assume_optically_thick=False, scale_height=H, verbose=False).fit(u, v, V, w)
    assert np.array_equal(a.I, b.I)
    a = FourierBesselDebrisFitter(2.0, 20, geom, H, verbose=False).fit(u, v, V, w)
    b = FourierBesselFitter(2.0, 20, geom, assume_optically_thick=False, scale_height=H, verbose=False).fit(u, v, V, w)
    assert np.array_equal(a.I, b.I)
    # the residuals of a debris profile: the per-column factor exp(-kz^2 H2[k]) inside fh_vis_residuals against predict
    from frank_amd import _lib
    from frank_amd.geometry import _ResidentTable
    t = _ResidentTable(0, u, v, V, w)
    fb = FourierBesselDebrisFitter(2.0, 20, geom, H, verbose=False)
    sol = fb.fit(u, v, V, w)
    out, gg = np.empty(2 * u.size), _lib.make_geometry(geom)
    _lib.check(_lib.lib.fh_vis_residuals(fb._DHT.context(), ctypes.byref(gg), 2, t.handle, 0, u.size, _lib.ptr(np.ascontiguousarray(sol.I)),
                                         _lib.ptr(out), None))
    e = np.sqrt(w) * (sol.predict(u, v) - V)
    assert np.abs(out - np.concatenate([e.real, e.imag])).max() < 1e-9 * np.abs(e).max()
    t.close()


def test_normal_equations_entry_points(golden):
    """J^T J, J^T r summed on the device against the same sums of the Jacobians copied out: the Gaussian's analytic one
    (fh_gauss_normal_equations vs fh_gauss_residuals) and the forward-difference one of residual vectors kept in slots
    (fh_residual_normal_equations vs differences of fh_vis_residuals outputs)."""
    from frank_amd import DiscreteHankelTransform, FixedGeometry, _lib
    from frank_amd._levmar import forward_steps
    from frank_amd.geometry import FitGeometryFourierBessel, _ResidentTable
    g = golden("geometry_fits_2e4.npz")
    u, v, V, w = table(g)
    t = _ResidentTable(0, u, v, V, w)
    x = np.array([0.6, 1.4, 0.003, -0.002, 0.8, 0.7])
    for fit_ip, fit_ph in ((1, 1), (0, 1), (1, 0)):
        fun, jac, ss = np.empty(2 * u.size), np.empty((2 * u.size, 6)), ctypes.c_double()
        _lib.check(_lib.lib.fh_gauss_residuals(t.handle, _lib.ptr(x), fit_ip, fit_ph, _lib.ptr(fun), _lib.ptr(jac), None))
        A, b = np.empty((6, 6)), np.empty(6)
        _lib.check(_lib.lib.fh_gauss_normal_equations(t.handle, _lib.ptr(x), fit_ip, fit_ph, _lib.ptr(A), _lib.ptr(b), ctypes.byref(ss)))
        JtJ, Jtf = jac.T @ jac, jac.T @ fun
        scale = np.sqrt(np.outer(np.diag(JtJ), np.diag(JtJ))) + 1e-300
        assert np.abs((A - JtJ) / scale).max() < 1e-12 and np.array_equal(A, A.T)
        assert np.abs(b - Jtf).max() <= 1e-12 * np.abs(jac * fun[:, None]).sum(axis=0).max()
        assert abs(ss.value / (fun @ fun) - 1) < 1e-13
    DHT = DiscreteHankelTransform(float(g["Rmax"]) / rad_to_arcsec, int(g["N"]))
    f = FitGeometryFourierBessel(float(g["Rmax"]), int(g["N"]), optimizer="scipy")
    x4 = np.array(g["trial"])
    h = forward_steps(x4)
    r0 = f._residual(x4, uvdata=(DHT, t))
    cols = []
    for k in range(4):
        xk = x4.copy()
        xk[k] += h[k]
        cols.append((f._residual(xk, uvdata=(DHT, t)) - r0) / h[k])
    J = np.stack(cols, axis=1)
    for slot, xk in enumerate([x4] + [x4 + h[k] * np.eye(4)[k] for k in range(4)]):
        gg, I = f._profile_under(FixedGeometry(*xk), DHT, t)
        _lib.check(_lib.lib.fh_vis_residuals_slot(DHT.context(), ctypes.byref(gg), 0, t.handle, _lib.ptr(I), slot + 2, None))
    A, b = np.empty((4, 4)), np.empty(4)
    _lib.check(_lib.lib.fh_residual_normal_equations(DHT.context(), t.handle, 2, 4, (ctypes.c_int * 4)(3, 4, 5, 6), _lib.ptr(h), _lib.ptr(A),
                                                     _lib.ptr(b)))
    JtJ, Jtr = J.T @ J, J.T @ r0
    assert np.abs((A - JtJ) / np.sqrt(np.outer(np.diag(JtJ), np.diag(JtJ)))).max() < 1e-11
    assert np.abs(b - Jtr).max() <= 1e-11 * np.abs(J * r0[:, None]).sum(axis=0).max()
    # two free parameters only
    A2, b2 = np.empty((2, 2)), np.empty(2)
    _lib.check(_lib.lib.fh_residual_normal_equations(DHT.context(), t.handle, 2, 2, (ctypes.c_int * 4)(5, 6, 0, 0), _lib.ptr(h[2:]), _lib.ptr(A2),
                                                     _lib.ptr(b2)))
    assert np.allclose(A2, A[2:, 2:], rtol=1e-13) and np.allclose(b2, b[2:], rtol=1e-13)
    assert _lib.lib.fh_residual_normal_equations(DHT.context(), t.handle, 2, 5, (ctypes.c_int * 4)(3, 4, 5, 6), _lib.ptr(h), _lib.ptr(A), _lib.ptr(b)) != 0
    t.close()


def test_mock_data_helpers_against_the_reference(golden):
    """utilities.generic_dht / make_mock_data / add_vis_noise / get_collocation_points / draw_bootstrap_sample
    (utilities.py:634-666, 923-1146) on the reference's inputs: transforms to 1e-12 of the largest value, the seeded noise
    draws and the collocation points exactly."""
    from frank_amd import FixedGeometry
    from frank_amd.utilities import (add_vis_noise, draw_bootstrap_sample, generic_dht, get_collocation_points,
                                     make_mock_data)
    g = golden("mockdata_helpers.npz")
    r, I, u, v, w, N = g["r"], g["I"], g["u"], g["v"], g["w"], int(g["N"])

    def close(a, b, tol=1e-12):
        return np.abs(a - b).max() <= tol * np.abs(b).max()
    grid, f = generic_dht(r, I, 2.0, N)
    assert np.array_equal(grid, g["fwd_grid"]) and close(f, g["fwd"])
    q = np.hypot(u, v)
    assert close(generic_dht(r, I, 2.0, N, grid=q, inc=40.0)[1], g["fwd_on_q"])
    grid, b = generic_dht(g["fwd_grid"], g["fwd"], 2.0, N, direction="backward")
    assert np.array_equal(grid, g["bwd_grid"]) and close(b, g["bwd"], 1e-11)
    assert close(generic_dht(g["fwd_grid"], g["fwd"], 2.0, N, direction="backward", grid=g["rr"], inc=40.0)[1], g["bwd_on_r"], 1e-11)
    with pytest.raises(AttributeError):
        generic_dht(r, I, direction="sideways")
    qq, V = make_mock_data(r, I, 2.0, u, v, N=N)
    assert np.array_equal(qq, g["mock_plain_q"]) and close(V, g["mock_plain"])
    geom = FixedGeometry(40.0, 70.0, 0.0, 0.0)
    qq, V = make_mock_data(r, I, 2.0, u, v, projection="deproject", geometry=geom, N=N, add_noise=True, weights=w, seed=17)
    assert close(qq, g["mock_deproj_q"], 1e-15) and close(V, g["mock_deproj"])
    qq, V = make_mock_data(r, I, 2.0, u, v, projection="reproject", geometry=geom, N=N)
    assert close(qq, g["mock_reproj_q"], 1e-15) and close(V, g["mock_reproj"])
    for bad in (dict(projection="sideways"), dict(projection="deproject"), dict(geometry=geom)):
        with pytest.raises(AttributeError):
            make_mock_data(r, I, 2.0, u, v, N=N, **bad)
    assert np.array_equal(add_vis_noise(g["noise_complex_in"], w, seed=5), g["noise_complex"])
    assert np.array_equal(add_vis_noise(g["mock_plain"], w, seed=5), g["noise_real"])
    assert np.array_equal(get_collocation_points(2.0, N), g["coll_r"])
    assert np.array_equal(get_collocation_points(2.0, N, direction="backward"), g["coll_q"])
    with pytest.raises(AttributeError):
        get_collocation_points(direction="sideways")
    np.random.seed(3)
    ub, vb, Vb, wb = draw_bootstrap_sample(u, v, V, w)
    np.random.seed(3)
    pick = np.random.randint(low=0, high=len(u), size=len(u))
    assert np.array_equal(ub, u[pick]) and np.array_equal(wb, w[pick]) and np.array_equal(Vb, V[pick])


@pytest.mark.parametrize("N,n,geom", [(20, 3000, (34.97, 85.76, 1.9e-3, 2.5e-3)), (40, 20011, (60.0, 10.0, -0.05, 0.02)), (8, 257, (0.0, 0.0, 0.0, 0.0)),
                                      (20, 1, (10.0, 20.0, 0.0, 0.0))])
def test_residual_functions_against_the_oracle(N, n, geom):
    """fh_vis_residuals (after its binning pass: through the bucket tables) and fh_gauss_residuals against the CPU oracle's
    restatements at other sizes, geometries and uneven weights."""
    from oracle import oracle as fo
    from frank_amd import DiscreteHankelTransform, _lib
    from frank_amd.geometry import FitGeometryFourierBessel, _ResidentTable
    # (baselines beyond the last collocation frequency: without them the prior-free system is ill conditioned and two correct
    #  solvers differ by cond(M) eps in the fitted visibilities -- 3e-5 with qmax = 8e5 at N = 20)
    u, v, V, w = mock_disc_visibilities(max(n, 64), seed=N, noise_seed=n, weight=1e4, qmax=2e6)
    u, v, V = u[:n], v[:n], V[:n]
    w = w[:n] * np.random.default_rng(n).uniform(0.25, 4.0, n)
    DHT = DiscreteHankelTransform(2.0 / rad_to_arcsec, N)
    t = _ResidentTable(DHT.device, u, v, V, w)
    if n > N:  # (one row cannot determine N coefficients: both sides then solve a singular system their own way)
        r = FitGeometryFourierBessel(2.0, N, optimizer="scipy")._residual(geom, uvdata=(DHT, t))
        ref = fo.fourier_bessel_residual(N, 2.0 / rad_to_arcsec, geom, u, v, V, w)
        assert np.abs(r - ref).max() < 1e-7 * np.abs(ref).max()
    x = np.array([geom[0] * deg_to_rad, geom[1] * deg_to_rad, geom[2], geom[3], 0.9, 0.6])
    fun, jac = np.empty(2 * n), np.empty((2 * n, 6))
    _lib.check(_lib.lib.fh_gauss_residuals(t.handle, _lib.ptr(x), 1, 1, _lib.ptr(fun), _lib.ptr(jac), None))
    fun_o, jac_o = fo.gaussian_residual_and_jacobian(x, u, v, V, w)
    assert np.abs(fun - fun_o).max() <= 1e-12 * np.abs(fun_o).max()
    assert np.all(np.abs(jac - jac_o).max(axis=0) <= 1e-12 * np.maximum(np.abs(jac_o).max(axis=0), 1e-300))
    t.close()


def test_geometry_fits_on_single_precision_tables_and_scalar_weights(golden):
    """float32 / complex64 arrays are stored as a 20-byte table and widened on the fly: the fits equal, bit for bit, those of
    the double table holding the widened values; a scalar weight equals the constant array."""
    from frank_amd.geometry import FitGeometryFourierBessel, FitGeometryGaussian
    g = golden("geometry_fits_2e4.npz")
    u, v, V, w = table(g)
    u4, v4, V4, w4 = u.astype(np.float32), v.astype(np.float32), V.astype(np.complex64), w.astype(np.float32)
    wide = (u4.astype(np.float64), v4.astype(np.float64), V4.astype(np.complex128), w4.astype(np.float64))
    for make, ref in ((lambda: FitGeometryGaussian(guess=[30.0, 80.0, 0.0, 0.0]), g["gauss_free"]),
                      (lambda: FitGeometryFourierBessel(2.0, 20, guess=[30.0, 80.0, 0.0, 0.0]), g["fb_free"])):
        a, b, c = make(), make(), make()
        a.fit(u4, v4, V4, w4)
        b.fit(*wide)
        c.fit(wide[0], wide[1], wide[2], float(w4[0]))
        assert np.array_equal(geometry_of(a), geometry_of(b))
        assert np.abs(geometry_of(c) - geometry_of(b)).max() < 1e-9
        assert np.abs(geometry_of(a)[:2] - ref[:2]).max() < 0.05  # (and they are fits: the fp64 table's within the rounding of the data)


@pytest.mark.parametrize("N", [300, 400])
def test_residual_pass_of_a_frank_fit_at_large_N(N):
    """fh_vis_residuals with the profile of a full FrankFitter fit (the chi^2 a caller computes after a fit, radial_fitters.py:56-98
    + io.py:213) against sqrt(w) (sol.predict(u, v) - V): N = 300 right after the fit's binning pass (bucket tables), N = 400 on
    the rows path of the wide basis (N Bessel evaluations per row)."""
    from frank_amd import FixedGeometry, FrankFitter, _lib
    from frank_amd.geometry import _ResidentTable
    n = 50000
    u, v, V, w = mock_disc_visibilities(n, seed=5, noise_seed=6)
    geom = FixedGeometry(34.97, 85.76, 1.9e-3, 2.5e-3)
    FF = FrankFitter(2.0, N, geom, verbose=False)
    sol = FF.fit(u, v, V, w)
    e = np.sqrt(w) * (sol.predict(u, v) - V)
    t = _ResidentTable(0, u, v, V, w)
    ctx, gg = FF._DHT.context(), _lib.make_geometry(geom)
    out, ss = np.empty(2 * n), ctypes.c_double()
    I = np.ascontiguousarray(sol.I)
    for bin_first in (False, True):
        if bin_first:  # the same rows under the same geometry binned just before: the table path where it exists
            _lib.check(_lib.lib.fh_bin_reset(ctx))
            _lib.check(_lib.lib.fh_bin_visibilities(ctx, ctypes.byref(gg), t.handle, 0, n))
        _lib.check(_lib.lib.fh_vis_residuals(ctx, ctypes.byref(gg), 0, t.handle, 0, n, _lib.ptr(I), _lib.ptr(out), ctypes.byref(ss)))
        assert np.abs(out - np.concatenate([e.real, e.imag])).max() < 1e-9 * np.abs(e).max(), bin_first
        assert abs(ss.value / np.sum(np.abs(e) ** 2) - 1) < 1e-10
    t.close()


def test_save_fit_and_load_sol(tmp_path, golden):
    """io.save_fit / load_sol (io.py:127-218) around a fit of the reference's real-data table: every file it names, the
    residual table = data - model, the pickled solution giving back the profile and predicting again."""
    from frank_amd import FixedGeometry, FrankFitter
    from frank_amd import io as fio
    from frank_amd.utilities import get_fit_stat_uncer
    g = golden("realdata_multi_ring_N100.npz")
    u, v, w = g["u"], g["v"], g["w"]
    V = g["Vre"] + 1j * g["Vim"]
    FF = FrankFitter(2.0, 100, FixedGeometry(0.0, 0.0), verbose=False, store_iteration_diagnostics=True)
    sol = FF.fit(u, v, V, w)
    assert np.abs(sol.I - g["I"]).max() < 1e-6 * np.abs(g["I"]).max()
    prefix = str(tmp_path / "disc")
    fio.save_fit(u, v, V, w, sol, prefix, save_iteration_diag=True, iteration_diag=FF.iteration_diagnostics, format="npz")
    for tail in ("_frank_sol.obj", "_frank_iteration_diagnostics.obj", "_frank_profile_fit.txt", "_frank_vis_fit.npz",
                 "_frank_uv_fit.npz", "_frank_uv_resid.npz"):
        assert (tmp_path / ("disc" + tail)).exists(), tail
    prof = np.loadtxt(prefix + "_frank_profile_fit.txt")
    assert np.array_equal(prof[:, 0], sol.r) and np.array_equal(prof[:, 1], sol.I)
    np.testing.assert_allclose(prof[:, 2], get_fit_stat_uncer(sol), rtol=1e-15)
    uf, vf, Vf, wf = fio.load_uvtable(prefix + "_frank_uv_fit.npz")
    ur, vr, Vr, wr = fio.load_uvtable(prefix + "_frank_uv_resid.npz")
    assert np.array_equal(uf, u) and np.array_equal(wr, w) and np.array_equal(Vr, V - Vf)
    assert np.abs(Vf - sol.predict(u, v)).max() == 0.0
    back = fio.load_sol(prefix + "_frank_sol.obj")
    assert np.array_equal(back.I, sol.I) and np.array_equal(back.predict(u[:100], v[:100]), Vf[:100])
    fio.save_fit(u, v, V, w, sol, prefix + "_t", save_solution=False, save_uvtables=False, format="txt")
    vis_fit = np.loadtxt(prefix + "_t_frank_vis_fit.txt")
    assert np.array_equal(vis_fit[:, 0], sol.q) and vis_fit.shape == (100, 2)


def test_multi_frequency_mapping(golden):
    """map_visibilities(..., frequencies) (statistical_models.py:175-237): one (M, j) per channel against the reference's,
    the channels, the flag, the single null likelihood; the channels add up to the single-channel mapping."""
    from frank_amd import DiscreteHankelTransform, FixedGeometry, VisibilityMapping
    from frank_amd.mock import MOCK_GEOMETRY
    g = golden("multifreq_N50_2e4.npz")
    u, v, V, w = mock_disc_visibilities(int(g["n"]), seed=int(g["seed"]), noise_seed=int(g["noise_seed"]))
    freq = g["channel_values"][np.random.default_rng(int(g["freq_seed"])).integers(0, 3, int(g["n"]))]
    assert hashlib.sha256(b"".join(np.ascontiguousarray(a).tobytes() for a in (u, v, V, w, freq))).hexdigest() == str(g["input_sha256"])
    vm = VisibilityMapping(DiscreteHankelTransform(2.0 / rad_to_arcsec, int(g["N"])), FixedGeometry(**MOCK_GEOMETRY), verbose=False)
    m = vm.map_visibilities(u, v, V, w, frequencies=freq)
    assert m["mult_freq"] is True and np.array_equal(m["channels"], g["channels"]) and m["hash"][0] is True
    assert m["M"].shape == (3, 50, 50) and m["j"].shape == (3, 50)
    for i in range(3):
        assert np.abs(m["M"][i] - g["M"][i]).max() < 5e-13 * np.abs(g["M"][i]).max(), i
        assert np.abs(m["j"][i] - g["j"][i]).max() < 5e-13 * np.abs(g["j"][i]).max(), i
    assert abs(m["null_likelihood"] / float(g["H0"]) - 1) < 1e-12
    one = vm.map_visibilities(u, v, V, w)
    assert one["mult_freq"] is False and np.abs(m["M"].sum(axis=0) - one["M"]).max() < 1e-13 * np.abs(one["M"]).max()
    assert vm.check_hash(m["hash"], multi_freq=True) and not vm.check_hash(m["hash"])
    with pytest.raises(ValueError):
        vm.map_visibilities(u, v, V, w, frequencies=freq[:-1])


def test_device_resident_operands_of_an_evaluation():
    """fh_gaussian_model with M = j = NULL solves the statistics fh_stats_finalize left on the device, fh_vis_residuals_slot with
    I = NULL takes that solve's profile: the same numbers as handing the arrays through the host; an error without statistics."""
    from frank_amd import DiscreteHankelTransform, FixedGeometry, _lib
    from frank_amd.geometry import _ResidentTable
    N, n = 20, 30000
    u, v, V, w = mock_disc_visibilities(n, seed=3, noise_seed=4, qmax=1.5e6)
    DHT = DiscreteHankelTransform(2.0 / rad_to_arcsec, N)
    ctx, t = DHT.context(), _ResidentTable(0, u, v, V, w)
    sv = ctypes.c_int()
    assert _lib.lib.fh_gaussian_model(ctx, None, None, None, None, None, None, ctypes.byref(sv)) != 0  # nothing binned yet
    assert _lib.lib.fh_gaussian_model(ctx, _lib.ptr(np.eye(N)), None, None, None, None, None, ctypes.byref(sv)) != 0
    g = _lib.make_geometry(FixedGeometry(30.0, 80.0, 0.01, -0.005))
    H0, a, b = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    M, j, I, I2 = np.empty((N, N)), np.empty(N), np.empty(N), np.empty(N)
    s1, s2 = ctypes.c_double(), ctypes.c_double()
    for host in (True, False):
        _lib.check(_lib.lib.fh_bin_reset(ctx))
        _lib.check(_lib.lib.fh_bin_visibilities(ctx, ctypes.byref(g), t.handle, 0, n))
        if host:
            _lib.check(_lib.lib.fh_stats_finalize(ctx, ctypes.byref(g), 0, 0, _lib.ptr(M), _lib.ptr(j), ctypes.byref(H0), ctypes.byref(a), ctypes.byref(b)))
            _lib.check(_lib.lib.fh_gaussian_model(ctx, _lib.ptr(M), _lib.ptr(j), None, _lib.ptr(I), None, None, ctypes.byref(sv)))
            _lib.check(_lib.lib.fh_vis_residuals_slot(ctx, ctypes.byref(g), 0, t.handle, _lib.ptr(I), 0, ctypes.byref(s1)))
        else:
            _lib.check(_lib.lib.fh_stats_finalize(ctx, ctypes.byref(g), 0, 0, None, None, ctypes.byref(H0), ctypes.byref(a), ctypes.byref(b)))
            _lib.check(_lib.lib.fh_gaussian_model(ctx, None, None, None, _lib.ptr(I2), None, None, ctypes.byref(sv)))
            _lib.check(_lib.lib.fh_vis_residuals_slot(ctx, ctypes.byref(g), 0, t.handle, None, 1, ctypes.byref(s2)))
    assert np.array_equal(I, I2) and s1.value == s2.value
    t.close()


def test_predict_sky_through_the_bucket_tables():
    """sol.predict(u, v) on a large call (one look at the baselines, then a polynomial per row) against the same call with N
    Bessel evaluations per row, and a small call (always direct) on a slice."""
    import os
    from frank_amd import FixedGeometry, FrankFitter
    n = 200000
    u, v, V, w = mock_disc_visibilities(n, seed=5, noise_seed=6)
    sol = FrankFitter(2.0, 120, FixedGeometry(34.97, 85.76, 1.9e-3, 2.5e-3), verbose=False).fit(u, v, V, w)
    P = sol.predict(u, v)
    os.environ["FRANK_AMD_RESIDUAL_DIRECT"] = "1"
    try:
        Pd = sol.predict(u, v)
    finally:
        del os.environ["FRANK_AMD_RESIDUAL_DIRECT"]
    assert 0 < np.abs(P - Pd).max() < 1e-11 * np.abs(Pd).max()
    assert np.array_equal(sol.predict(u[:5000], v[:5000]), Pd[:5000])
    # the fit's own statistics are untouched by the call: binning the same table again gives the same fit
    again = FrankFitter(2.0, 120, FixedGeometry(34.97, 85.76, 1.9e-3, 2.5e-3), verbose=False).fit(u, v, V, w)
    assert np.array_equal(again.I, sol.I)


def test_fourier_bessel_interpolation_against_the_reference(golden):
    """DiscreteHankelTransform.interpolation_coefficients / interpolate, VisibilityMapping.interpolate and
    FrankRadialFit.interpolate_brightness (hankel.py:206-263, statistical_models.py:435-481, radial_fitters.py:146-176)
    against the reference's own outputs (tools/make_golden_interp.py): both spaces, points beyond Rmax / Qmax (the series is
    cut there), r = 0, a point between two collocation points, an array of any shape in chunks; and the gaussian of the
    reference's test_hankel_gauss between its collocation points (tests.py:37-82: atol 1e-4 for generic points)."""
    from frank_amd import DiscreteHankelTransform, FixedGeometry, VisibilityMapping
    from frank_amd.constants import rad_to_arcsec
    g = golden("interpolate.npz")
    for N, Rmax in ((100, 5.0), (300, 2.0 / rad_to_arcsec)):
        d = DiscreteHankelTransform(Rmax, N)
        for pts, space, Y, fin, fout in (("rpts", "Real", "Yreal", "f", "freal"), ("qpts", "Fourier", "Yfourier", "g", "gfourier")):
            Yd = d.interpolation_coefficients(g["N%d_%s" % (N, pts)], space)
            Yr = g["N%d_%s" % (N, Y)]
            assert Yd.shape == Yr.shape
            assert np.abs(Yd - Yr).max() <= 1e-11 * np.abs(Yr).max()
            out = d.interpolate(g["N%d_%s" % (N, fin)], g["N%d_%s" % (N, pts)], space)
            np.testing.assert_allclose(out, g["N%d_%s" % (N, fout)], rtol=0, atol=1e-11 * np.abs(g["N%d_%s" % (N, fout)]).max())
        with pytest.raises(ValueError):
            d.interpolation_coefficients(np.array([0.1]), "sideways")    # hankel.py:229-231
    d = DiscreteHankelTransform(5.0, 100)
    out = d.interpolate(np.exp(-0.5 * d.r ** 2), g["gauss_r"], "Real")
    np.testing.assert_allclose(out, g["gauss_interp"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(out[:-1], np.exp(-0.5 * g["gauss_r"][:-1] ** 2), rtol=0, atol=1e-4)   # (r = Rmax: the series is 0)
    d = DiscreteHankelTransform(2.0 / rad_to_arcsec, 50)
    vm = VisibilityMapping(d, FixedGeometry(30., 40., 0., 0.), block_size=700, verbose=False)
    o = vm.interpolate(g["vm_I"], g["vm_R"], space="Real")
    assert o.shape == g["vm_out"].shape
    np.testing.assert_allclose(o, g["vm_out"], rtol=0, atol=1e-12 * np.abs(g["vm_out"]).max())
    # FrankRadialFit.interpolate_brightness: the MAP when no profile is given
    from frank_amd import FrankFitter
    u, v, V, w = mock_disc_visibilities(20000, seed=21, noise_seed=22)
    from frank_amd.mock import MOCK_GEOMETRY
    FF = FrankFitter(2.0, 60, FixedGeometry(**MOCK_GEOMETRY), alpha=1.3, weights_smooth=1e-2, verbose=False)
    sol = FF.fit(u, v, V, w)
    Rp = np.array([0.0, 0.31, 0.77, 1.5])
    a, b = sol.interpolate_brightness(Rp), sol.interpolate_brightness(Rp, sol.I)
    assert np.array_equal(a, b) and a.shape == Rp.shape
    # (AT a collocation point the reference's formula is 0 / 0; a hair beside it the series returns the point's value)
    np.testing.assert_allclose(sol.interpolate_brightness(sol.r[5:8] * (1 + 1e-7)), sol.I[5:8], rtol=1e-4)

"""ctypes front-end of the CPU oracle (oracle/frank_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py -- never by anything under frank_amd/.
Every function names the reference lines it restates (see the C file).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libfrank_oracle.so")

FO_OK = 0
FO_ERR_QRANGE = -2
FO_ERR_BAD_P = -3
FO_ERR_NOT_SPD = -4


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("frank_oracle.c", "frank_oracle_lognormal.c")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libfrank_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _SO


_lib = None
_dp = ctypes.POINTER(ctypes.c_double)


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.fo_j0.restype = ctypes.c_double
        _lib.fo_j0.argtypes = [ctypes.c_double]
        _lib.fo_j1.restype = ctypes.c_double
        _lib.fo_j1.argtypes = [ctypes.c_double]
    return _lib


def _f8(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def j0(x):
    x = _f8(np.atleast_1d(x))
    out = np.empty_like(x)
    lib().fo_j0_array(_p(x), ctypes.c_int64(x.size), _p(out))
    return out


def jn_zeros0(nt):
    z = np.empty(nt)
    lib().fo_jn_zeros0(ctypes.c_int(nt), _p(z))
    return z


class DHT:
    """DiscreteHankelTransform.__init__ (hankel.py:55-93), nu = 0."""

    def __init__(self, Rmax, N):
        self.Rmax, self.N = float(Rmax), int(N)
        self.r, self.q, self.j_nk = np.empty(N), np.empty(N), np.empty(N)
        self.Ykm, self.scale_factor = np.empty((N, N)), np.empty(N)
        jn, qm = ctypes.c_double(), ctypes.c_double()
        lib().fo_dht_setup(ctypes.c_double(Rmax), ctypes.c_int(N), _p(self.r), _p(self.q), _p(self.j_nk),
                           ctypes.byref(jn), ctypes.byref(qm), _p(self.Ykm), _p(self.scale_factor))
        self.j_nN, self.Qmax = jn.value, qm.value

    def coefficients(self, q=None):
        """hankel.py:187-204 (forward)."""
        N = self.N
        if q is None:
            Y = np.empty((N, N))
            lib().fo_dht_coefficients_self(ctypes.c_int(N), ctypes.c_double(self.j_nN),
                                           ctypes.c_double(self.Qmax), _p(self.Ykm), _p(Y))
            return Y
        q = _f8(q)
        H = np.empty((q.size, N))
        lib().fo_dht_coefficients(ctypes.c_int(N), ctypes.c_double(self.Qmax), _p(self.j_nk),
                                  _p(self.scale_factor), _p(q), ctypes.c_int64(q.size), _p(H))
        return H

    def transform(self, f):
        """hankel.py:151-165 (forward, q=None)."""
        f = _f8(f)
        out = np.empty(self.N)
        lib().fo_dht_transform_forward(ctypes.c_int(self.N), ctypes.c_double(self.Rmax),
                                       ctypes.c_double(self.j_nN), _p(self.Ykm), _p(f), _p(out))
        return out


def apply_correction(u, v, V, inc, PA, dRA, dDec):
    """SourceGeometry.apply_correction(use3D=True) (geometry.py:202-236)."""
    u, v = _f8(u), _f8(v)
    V = np.asarray(V)
    Vre = _f8(V.real)
    Vim = _f8(V.imag) if np.iscomplexobj(V) else None
    n = u.size
    up, vp, wp, Vr, Vi = (np.empty(n) for _ in range(5))
    lib().fo_apply_correction(ctypes.c_int64(n), _p(u), _p(v), _p(Vre), _p(Vim), ctypes.c_double(inc),
                              ctypes.c_double(PA), ctypes.c_double(dRA), ctypes.c_double(dDec), _p(up), _p(vp),
                              _p(wp), _p(Vr), _p(Vi))
    return up, vp, wp, Vr + 1j * Vi


def map_visibilities(N, Rmax, geom, u, v, V, w, vis_model=0, check_qbounds=True, block_size=10 ** 5, H2=None):
    """VisibilityMapping.map_visibilities (statistical_models.py:109-237), single channel.

    geom = (inc_deg, PA_deg, dRA_arcsec, dDec_arcsec); Rmax in radians.  vis_model 2 ('debris') needs
    H2 = 0.5 * (2 pi scale_height(r) / rad_to_arcsec)**2 (:101-102).
    Returns dict(M, j, null_likelihood, qmin, qmax, rc).
    """
    u, v = _f8(u), _f8(v)
    V = np.asarray(V)
    Vre = _f8(V.real)
    Vim = _f8(V.imag) if np.iscomplexobj(V) else None
    w = _f8(np.atleast_1d(w))
    n = u.size
    M, j = np.zeros((N, N)), np.zeros(N)
    H0, qmin, qmax = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    rc = lib().fo_map_visibilities_ex(
        ctypes.c_int(N), ctypes.c_double(Rmax), ctypes.c_double(geom[0]), ctypes.c_double(geom[1]),
        ctypes.c_double(geom[2]), ctypes.c_double(geom[3]), ctypes.c_int(vis_model),
        ctypes.c_int(1 if check_qbounds else 0), ctypes.c_int64(block_size), ctypes.c_int64(n), _p(u), _p(v),
        _p(Vre), _p(Vim), _p(w), ctypes.c_int64(w.size), _p(None if H2 is None else _f8(H2)), _p(M), _p(j),
        ctypes.byref(H0), ctypes.byref(qmin), ctypes.byref(qmax))
    return dict(M=M, j=j, null_likelihood=H0.value, qmin=qmin.value, qmax=qmax.value, rc=rc)


def gaussian_model(dht, M, j, p=None):
    """GaussianModel(DHT, M, j, p) (statistical_models.py:650-760). Returns (mu, chol_upper, Sinv, rc)."""
    N = dht.N
    Y = dht.coefficients()
    M, j = _f8(M), _f8(j)
    mu, chol, Sinv = np.empty(N), np.empty((N, N)), np.zeros((N, N))
    pp = None if p is None else _f8(p)
    rc = lib().fo_gaussian_model(ctypes.c_int(N), _p(Y), _p(M), _p(j), _p(pp), _p(mu), _p(chol), _p(Sinv))
    return mu, chol, Sinv, rc


def smoothing_matrix(dht, weights):
    """spectral_smoothing_matrix (filter.py:23-62) as a dense N x N array."""
    N = dht.N
    band = np.empty((5, N))
    lib().fo_smoothing_matrix(ctypes.c_int(N), _p(dht.q), ctypes.c_double(weights), _p(band))
    T = np.zeros((N, N))
    for d in range(-2, 3):
        for i in range(N):
            if 0 <= i + d < N:
                T[i, i + d] = band[d + 2, i]
    return T, band


def update_power_spectrum(dht, band, alpha, p0, p, mu, chol):
    """CriticalFilter.update_power_spectrum (filter.py:154-177)."""
    N = dht.N
    Y = dht.coefficients()
    out = np.empty(N)
    lib().fo_update_power_spectrum(ctypes.c_int(N), _p(Y), _p(_f8(band)), ctypes.c_double(alpha),
                                   ctypes.c_double(p0), _p(_f8(p)), _p(_f8(mu)), _p(_f8(chol)), _p(out))
    return out


def frank_fit_normal(N, Rmax, M, j, alpha=1.05, p0=1e-15, wsmooth=1e-4, tol=1e-3, max_iter=2000,
                     diagnostics=False):
    """FrankFitter._fit, method='Normal' (radial_fitters.py:737-832). Rmax in radians.

    Returns dict(mu, p, niter, rc[, diag_p, diag_mu]).
    """
    M, j = _f8(M), _f8(j)
    mu, p = np.empty(N), np.empty(N)
    niter, nsvd = ctypes.c_int(0), ctypes.c_int(0)
    dp = dm = None
    if diagnostics:
        dp, dm = np.zeros((max_iter + 1, N)), np.zeros((max_iter + 1, N))
    rc = lib().fo_frank_fit_normal(ctypes.c_int(N), ctypes.c_double(Rmax), _p(M), _p(j), ctypes.c_double(alpha),
                                   ctypes.c_double(p0), ctypes.c_double(wsmooth), ctypes.c_double(tol),
                                   ctypes.c_int(max_iter), _p(mu), _p(p), ctypes.byref(niter), _p(dp), _p(dm),
                                   ctypes.byref(nsvd))
    out = dict(mu=mu, p=p, niter=niter.value, rc=rc, n_svd=nsvd.value)
    if diagnostics:
        out["diag_p"], out["diag_mu"] = dp[:niter.value], dm[:niter.value]
    return out


def lognormal_map(dht, M, j, p, guess, s0):
    """LogNormalMAPModel(DHT, M, j, p, guess=guess, s0=s0) (statistical_models.py:1012-1160).

    Returns dict(s, Dinv, chol, Sinv, rc, stats=(status, nstep, nfev, nhess))."""
    N = dht.N
    Y = dht.coefficients()
    s = np.array(guess, dtype="f8", order="C")
    Dinv, chol, Sinv = np.empty((N, N)), np.empty((N, N)), np.empty((N, N))
    stats = (ctypes.c_long * 4)()
    rc = lib().fo_lognormal_map(ctypes.c_int(N), _p(Y), _p(_f8(M)), _p(_f8(j)), _p(_f8(p)), ctypes.c_double(s0),
                                _p(s), _p(Dinv), _p(chol), _p(Sinv), stats)
    return dict(s=s, Dinv=Dinv, chol=chol, Sinv=Sinv, rc=rc, stats=tuple(stats))


def frank_fit_lognormal(N, Rmax, M, j, alpha=1.05, p0=1e-35, wsmooth=1e-4, tol=1e-3, max_iter=2000, I_scale=1e5,
                        diagnostics=False):
    """FrankFitter._fit, method='LogNormal' (radial_fitters.py:737-832). Rmax in radians.

    Returns dict(s, I, p, niter, rc, Dinv, totals=(fits, nstep, nfev, nhess), status_hist[, diag_p, diag_s])."""
    M, j = _f8(M), _f8(j)
    s, p, Dinv = np.empty(N), np.empty(N), np.empty((N, N))
    niter = ctypes.c_int(0)
    totals, hist = (ctypes.c_long * 4)(), (ctypes.c_long * 5)()
    dp = ds = None
    if diagnostics:
        dp, ds = np.zeros((max_iter + 1, N)), np.zeros((max_iter + 1, N))
    s0 = float(np.log(I_scale))
    rc = lib().fo_frank_fit_lognormal(ctypes.c_int(N), ctypes.c_double(Rmax), _p(M), _p(j), ctypes.c_double(alpha),
                                      ctypes.c_double(p0), ctypes.c_double(wsmooth), ctypes.c_double(tol),
                                      ctypes.c_int(max_iter), ctypes.c_double(s0), _p(s), _p(p), ctypes.byref(niter),
                                      _p(Dinv), _p(dp), _p(ds), totals, hist)
    out = dict(s=s, I=np.exp(s + s0), p=p, niter=niter.value, rc=rc, Dinv=Dinv, totals=tuple(totals),
               status_hist=tuple(hist))
    if diagnostics:
        out["diag_p"], out["diag_s"] = dp[:niter.value], ds[:niter.value]
    return out


def uvbin_nbins(uv, bin_width):
    uv = _f8(uv)
    f = lib().fo_uvbin_nbins
    f.restype = ctypes.c_int64
    return int(f(_p(uv), ctypes.c_int64(uv.size), ctypes.c_double(bin_width)))


def uvbin_determine(uv, bin_width, nbins):
    """UVDataBinner.determine_uv_bin (utilities.py:271-298)."""
    uv = _f8(uv)
    out = np.empty(uv.size, dtype=np.int32)
    lib().fo_uvbin_determine(_p(uv), ctypes.c_int64(uv.size), ctypes.c_double(bin_width), ctypes.c_int64(nbins),
                             out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return out


def uvbin_build(uv, V, w, bin_width):
    """UVDataBinner.__init__ (utilities.py:203-268). Returns dict(nbins, uv, V, w, count, err) with NaN-filled empties."""
    uv, w = _f8(uv), _f8(w)
    V = np.asarray(V)
    cplx = np.iscomplexobj(V)
    Vre = _f8(V.real)
    Vim = _f8(V.imag) if cplx else None
    nb = uvbin_nbins(uv, bin_width)
    buv, bre, bim, bw = np.zeros(nb), np.zeros(nb), np.zeros(nb), np.zeros(nb)
    ere, eim = np.zeros(nb), np.zeros(nb)
    cnt = np.zeros(nb, dtype=np.int64)
    lib().fo_uvbin_build(_p(uv), _p(Vre), _p(Vim), _p(w), ctypes.c_int64(uv.size), ctypes.c_double(bin_width),
                         ctypes.c_int64(nb), _p(buv), _p(bre), _p(bim), _p(bw),
                         cnt.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _p(ere), _p(eim))
    return dict(nbins=nb, uv=buv, V=(bre + 1j * bim) if cplx else bre, w=bw, count=cnt,
                err=(ere + 1j * eim) if cplx else ere)


# ---- geometry fits: the residual functions the optimiser sees (geometry.py:404-763) --------------------------------------
def fourier_bessel_residual(N, Rmax, geom, u, v, V, w):
    """FitGeometryFourierBessel._residual (geometry.py:660-694): the prior-free fit under `geom` = (inc, PA [deg], dRA,
    dDec [arcsec]) -- map_visibilities, M I = j -- then w**0.5 * (sol.predict(u, v) - V), real parts before imaginary
    parts.  predict (radial_fitters.py:56-98): deproject, H(q) I cos(inc), rotate by the phase centre.  Rmax in radians."""
    inc, PA, dRA, dDec = geom
    u, v, V = _f8(u), _f8(v), np.asarray(V, dtype=np.complex128)
    w = np.broadcast_to(_f8(np.atleast_1d(w)), u.shape)
    m = map_visibilities(N, Rmax, geom, u, v, V, w, vis_model=0, check_qbounds=False)
    dht = DHT(Rmax, N)
    I = gaussian_model(dht, m["M"], m["j"], None)[0]
    up, vp, _, _ = apply_correction(u, v, V, inc, PA, dRA, dDec)
    Vm = (dht.coefficients(np.hypot(up, vp)) * np.cos(inc * np.pi / 180.0)) @ I
    phi = (u * dRA + v * dDec) * (2.0 * np.pi / (3600.0 * 180.0 / np.pi))
    e = np.sqrt(w) * (Vm * (np.cos(phi) + 1j * np.sin(phi)) - V)
    return np.concatenate([e.real, e.imag])


def gaussian_residual_and_jacobian(x, u, v, V, w, fit_inc_pa=True, fit_phase=True):
    """_gauss_fun and _gauss_jac of _fit_geometry_gaussian (geometry.py:535-585) at x = (inc, PA [rad], dRA, dDec [arcsec],
    norm, scal): the residual (real parts, then imaginary parts) and its [2 n][6] Jacobian."""
    r2a = 3600.0 * 180.0 / np.pi
    fac, sw = 2 * np.pi / r2a, np.sqrt(np.broadcast_to(_f8(np.atleast_1d(w)), np.shape(u)))
    inc, PA, dRA, dDec, norm, scal = x

    def wrap(z):
        z = np.asarray(z, dtype=np.complex128)
        return np.concatenate([z.real, z.imag])
    phi = dRA * fac * u + dDec * fac * v
    Vp = V * (np.cos(phi) - 1j * np.sin(phi))
    c_t, s_t, c_i, s_i = np.cos(PA), np.sin(PA), np.cos(inc), np.sin(inc)
    up, vp = u * c_t - v * s_t, u * s_t + v * c_t
    uv = up * up * c_i * c_i + vp * vp
    G = sw * np.exp(-0.5 * uv / (scal * r2a) ** 2)
    fun = wrap(norm * G - sw * Vp)
    jac = np.zeros((6, 2 * len(sw)))
    nn = norm / (scal * r2a) ** 2
    if fit_phase:
        dVp = -sw * V * (-np.sin(phi) - 1j * np.cos(phi)) * fac
        jac[2], jac[3] = wrap(dVp * u), wrap(dVp * v)
    if fit_inc_pa:
        jac[0] = wrap(nn * G * up * up * c_i * s_i)
        jac[1] = wrap(nn * G * up * vp * (c_i * c_i - 1) / 2)
    jac[4] = wrap(G)
    jac[5] = wrap(nn * G * uv / scal)
    return fun, jac.T

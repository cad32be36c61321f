"""Source geometry (inclination, position angle, phase centre) for the MI355X frank path.

API-compatible with frank/geometry.py: `apply_phase_shift` :41-79, `deproject` :82-131, `SourceGeometry` :173-369,
`FixedGeometry` :372-401, and the two geometry fits, `FitGeometryGaussian` :404-497 and `FitGeometryFourierBessel`
:600-763.  The hot path does NOT run the NumPy routines at the top of this file: `VisibilityMapping.map_visibilities`
hands the raw (u, v, V) to the GPU, where the pre-pass applies the phase shift and the deprojection.  They serve callers
that need deprojected coordinates on the host (`FrankRadialFit.predict`, user code) and carry the (inc, PA, dRA, dDec)
that the kernels read.

The geometry fits are CALLERS of the hot path: the reference hands a residual function over the whole table to
scipy.optimize.least_squares(method='lm'), and for the non-parametric fit every evaluation of it is a binning pass, a
solve and a prediction under a trial geometry.  Here the table is uploaded once and stays in HBM; a residual evaluation is
fh_bin_visibilities + fh_gaussian_model + fh_vis_residuals (csrc/vis_residual.hip) under the trial geometry, or
fh_gauss_residuals for the Gaussian.  The optimiser: by default (`optimizer='device'`) MINPACK's Levenberg-Marquardt
algorithm on normal equations summed on the device (frank_amd/_levmar.py) -- over the FREE parameters only, where the
reference keeps pinned ones in x with zero Jacobian columns: xnorm, the first trust radius and the `delta <= xtol xnorm`
test see a shorter vector, and function evaluations are counted differently (same geometry to the tolerances of the tests,
a different path); `optimizer='scipy'` is the reference's routine with the reference's arguments.
"""
import ctypes
import logging

import numpy as np

from frank_amd.constants import rad_to_arcsec, deg_to_rad

_TWO_PI_PER_ARCSEC = 2. * np.pi / rad_to_arcsec


def _phasor(u, v, dRA, dDec):
    """exp(i * 2 pi (u dRA + v dDec)) with the offsets given in arcsec."""
    angle = u * (dRA * _TWO_PI_PER_ARCSEC) + v * (dDec * _TWO_PI_PER_ARCSEC)
    return np.cos(angle) + 1j * np.sin(angle)


def apply_phase_shift(u, v, V, dRA, dDec, inverse=False):
    """Move the source by (dRA, dDec) arcsec in the image plane; `inverse=True` moves it back."""
    rot = _phasor(u, v, dRA, dDec)
    return V / rot if inverse else V * rot


def deproject(u, v, inc, PA, inverse=False):
    """Rotate the uv-plane by PA and compress u by cos(inc) (deproject), or undo that (`inverse=True`).

    Deprojecting also returns the third Fourier coordinate w' = u_rot * sin(inc).
    """
    ci, si = np.cos(inc * deg_to_rad), np.sin(inc * deg_to_rad)
    ct, st = np.cos(PA * deg_to_rad), np.sin(PA * deg_to_rad)
    if inverse:
        u = u / ci
        st = -st
        return u * ct - v * st, u * st + v * ct
    ur = u * ct - v * st
    vr = u * st + v * ct
    return ur * ci, vr, ur * si


class SourceGeometry(object):
    """Geometry container + correction helpers; inc, PA in degrees, dRA, dDec in arcsec."""

    def __init__(self, inc=None, PA=None, dRA=None, dDec=None):
        self._inc, self._PA, self._dRA, self._dDec = inc, PA, dRA, dDec

    # -- corrections -------------------------------------------------------------------------------------
    def apply_correction(self, u, v, V, use3D=False):
        """Centre the phase and deproject: returns (u', v'[, w'], V')."""
        Vc = apply_phase_shift(u, v, V, self._dRA, self._dDec, inverse=True)
        ud, vd, wd = deproject(u, v, self._inc, self._PA)
        return (ud, vd, wd, Vc) if use3D else (ud, vd, Vc)

    def undo_correction(self, u, v, V):
        """Reproject deprojected points and put the phase offset back."""
        us, vs = self.reproject(u, v)
        return us, vs, apply_phase_shift(us, vs, V, self._dRA, self._dDec)

    def deproject(self, u, v, use3D=False):
        out = deproject(u, v, self._inc, self._PA)
        return out if use3D else out[:2]

    def reproject(self, u, v):
        return deproject(u, v, self._inc, self._PA, inverse=True)

    def fit(self, u, v, V, weights):
        """Nothing to determine for a fixed geometry."""
        return None

    def clone(self):
        return FixedGeometry(self._inc, self._PA, self._dRA, self._dDec)

    # -- parameters ----------------------------------------------------------------------------------------
    inc = property(lambda self: self._inc, doc="inclination [deg]")
    PA = property(lambda self: self._PA, doc="position angle, east of north [deg]")
    dRA = property(lambda self: self._dRA, doc="phase-centre offset in right ascension [arcsec]")
    dDec = property(lambda self: self._dDec, doc="phase-centre offset in declination [arcsec]")

    @property
    def rescale_factor(self):
        """1 / cos(inc): the optically-thick flux rescaling."""
        return 1.0 / np.cos(self._inc * deg_to_rad)

    def __repr__(self):
        return "%s(inc=%r, PA=%r, dRA=%r, dDec=%r)" % (type(self).__name__, self._inc, self._PA, self._dRA, self._dDec)


class FixedGeometry(SourceGeometry):
    """Known geometry: FixedGeometry(inc, PA, dRA=0, dDec=0)."""

    def __init__(self, inc, PA, dRA=0.0, dDec=0.0):
        SourceGeometry.__init__(self, inc, PA, dRA, dDec)


_OPTIMIZERS = ('device', 'scipy')


def _fix_inc_and_PA_ranges(inc, PA):
    """Fold a fitted inclination into [0, 90] and a position angle into [0, 180) degrees (geometry.py:33-39)."""
    inc, PA = inc % 180, PA % 180
    return (180 - inc if inc > 90 else inc), PA


class _ResidentTable(object):
    """The (u, v, V, weights) of a geometry fit in HBM for as long as the optimiser runs (fh_vis_upload)."""

    def __init__(self, device, u, v, V, weights):
        from frank_amd import _lib
        self.handle = None  # (close() / __del__ must work on an object whose construction failed)
        self._lib = _lib
        V = np.asarray(V)
        # single-precision arrays are stored as they are (20 B per visibility) and widened as the kernels read them
        f32 = _lib.all_float32(u, v, V, weights)
        conv, ptr, upload = (_lib.f4, _lib.fptr, _lib.lib.fh_vis_upload_f32) if f32 else (_lib.f8, _lib.ptr, _lib.lib.fh_vis_upload)
        u, v = conv(u), conv(v)
        Vre, Vim = conv(V.real), (conv(V.imag) if np.iscomplexobj(V) else None)
        w = conv(np.atleast_1d(weights))
        self.n = u.size
        if v.size != self.n or Vre.size != self.n or w.size not in (1, self.n):
            raise ValueError("u, v, V (and weights) must have matching lengths")
        self.handle = ctypes.c_void_p()
        _lib.check(upload(int(device), ptr(u), ptr(v), ptr(Vre), ptr(Vim), ptr(w), w.size, self.n, ctypes.byref(self.handle)))

    def close(self):
        if self.handle is not None and self.handle.value:
            self._lib.lib.fh_vis_destroy(self.handle)
        self.handle = None

    __del__ = close


class FitGeometryGaussian(SourceGeometry):
    """Determine the geometry by fitting a Gaussian to the visibilities in the uv-plane (geometry.py:404-497).

    inc_pa = (inc, PA) [deg] and / or phase_centre = (dRA, dDec) [arcsec] fix those two instead of fitting them;
    guess = [inc, PA, dRA, dDec] starts the fit (default 10, 10, 0, 0).  Not in the reference: `device`, the HIP device,
    and `optimizer` -- 'device' (default): Levenberg-Marquardt on the normal equations, J^T J and J^T r summed on the GPU
    from the resident table by one streaming kernel per step (frank_amd/_levmar.py: MINPACK's algorithm, nothing of the
    table's size reaches the host); 'scipy': residuals and the 6-column Jacobian formed on the GPU, copied out and handed
    to scipy.optimize.least_squares(method='lm') exactly as the reference does.
    """

    def __init__(self, inc_pa=None, phase_centre=None, guess=None, device=None, optimizer='device'):
        super(FitGeometryGaussian, self).__init__()
        if optimizer not in _OPTIMIZERS:
            raise ValueError("optimizer must be one of %r, not %r" % (_OPTIMIZERS, optimizer))
        self._inc_pa, self._phase_centre, self._device, self._optimizer = inc_pa, phase_centre, device, optimizer
        guess = [10.0, 10.0, 0.0, 0.0] if guess is None else list(guess)
        guess = guess + [1.0, 1.0]  # normalisation and width of the Gaussian always start at one
        if inc_pa is not None:
            guess[0], guess[1] = inc_pa
        if phase_centre is not None:
            guess[2], guess[3] = phase_centre
        self._guess = guess

    def fit(self, u, v, V, weights):
        if self._inc_pa and self._phase_centre:
            logging.info('    You requested a Gaussian fit to determine the geometry, but you provided values for '
                         'inclination, PA, and the phase offset. --> Using your provided values (not fitting for the '
                         'geometry)')
            self._inc, self._PA = self._inc_pa
            self._dRA, self._dDec = self._phase_centre
            return
        logging.info('    Fitting Gaussian to determine geometry' +
                     (' (not fitting for inc or PA)' if self._inc_pa else
                      ' (not fitting for phase center)' if self._phase_centre else ''))
        inc, PA, dRA, dDec = _fit_geometry_gaussian(u, v, V, weights, self._guess, self._inc_pa, self._phase_centre,
                                                    device=self._device, optimizer=self._optimizer)
        if not self._inc_pa:
            inc, PA = _fix_inc_and_PA_ranges(inc, PA)
        self._inc, self._PA, self._dRA, self._dDec = inc, PA, dRA, dDec


def _fit_geometry_gaussian(u, v, V, weights, guess, inc_pa=None, phase_centre=None, device=None, optimizer='device'):
    """(inc, PA, dRA, dDec) of the best Gaussian, `guess` = [inc, PA (deg), dRA, dDec (arcsec), norm, width]
    (geometry.py:498-599)."""
    from frank_amd import _lib
    from frank_amd.hankel import default_device
    x0 = np.array(guess, dtype=np.float64)
    x0[:2] *= deg_to_rad
    if inc_pa is not None:
        x0[0], x0[1] = inc_pa[0] * deg_to_rad, inc_pa[1] * deg_to_rad
    if phase_centre is not None:
        x0[2], x0[3] = phase_centre
    table = _ResidentTable(default_device() if device is None else device, u, v, V, weights)
    fit_ip, fit_ph = int(inc_pa is None), int(phase_centre is None)
    pinned = x0.copy()

    def params_of(x):
        # a given pair never moves: the optimiser's value for it is ignored, as the reference's closures ignore it
        x = np.array(x, dtype=np.float64)
        if not fit_ip:
            x[:2] = pinned[:2]
        if not fit_ph:
            x[2:4] = pinned[2:4]
        return x

    def fun(x):
        out = np.empty(2 * table.n)
        _lib.check(_lib.lib.fh_gauss_residuals(table.handle, _lib.ptr(params_of(x)), fit_ip, fit_ph, _lib.ptr(out), None, None))
        return out

    def jac(x):
        out = np.empty((2 * table.n, 6))
        _lib.check(_lib.lib.fh_gauss_residuals(table.handle, _lib.ptr(params_of(x)), fit_ip, fit_ph, None, _lib.ptr(out), None))
        return out

    def device_fit():
        from frank_amd._levmar import levenberg_marquardt
        free = np.array([k for k in range(6) if (k >= 4 or (k < 2 and fit_ip) or (2 <= k < 4 and fit_ph))])

        def full(xf):
            x = pinned.copy()
            x[free] = xf
            return x

        def trial(xf):
            ss = ctypes.c_double()
            _lib.check(_lib.lib.fh_gauss_residuals(table.handle, _lib.ptr(full(xf)), fit_ip, fit_ph, None, None, ctypes.byref(ss)))
            return ss.value

        def normal(xf):
            A, g = np.empty((6, 6)), np.empty(6)
            _lib.check(_lib.lib.fh_gauss_normal_equations(table.handle, _lib.ptr(full(xf)), fit_ip, fit_ph, _lib.ptr(A), _lib.ptr(g), None))
            return A[np.ix_(free, free)], g[free], 0
        xf, info, _ = levenberg_marquardt(trial, lambda: None, normal, x0[free], maxfev=100 * 6)  # (least_squares with a callable jac: 100 n, n = 6)
        return full(xf)

    try:
        xbest = device_fit() if optimizer == 'device' else _lib.require_scipy(
            "optimizer='scipy' of the geometry fits").least_squares(fun, x0, jac=jac, method='lm').x
    finally:
        table.close()
    inc, PA, dRA, dDec = xbest[:4]
    inc, PA = (inc_pa if inc_pa is not None else (inc / deg_to_rad, PA / deg_to_rad))
    if phase_centre is not None:
        dRA, dDec = phase_centre
    return inc, PA, dRA, dDec


class FitGeometryFourierBessel(SourceGeometry):
    """Determine the geometry by minimising the chi^2 of a non-parametric (Fourier-Bessel, no prior) fit of the
    visibilities (geometry.py:600-763): FitGeometryFourierBessel(Rmax [arcsec], N, inc_pa=None, phase_centre=None,
    guess=None, verbose=False).  A small N keeps the prior-free fit stable.

    Every evaluation of the residual is, on the resident table: one binning pass under the trial geometry, the solve
    M I = j, and sqrt(w) (predict(u, v) - V) -- three C calls, the last one csrc/vis_residual.hip.
    Not in the reference: `device`, the HIP device, and `optimizer` -- 'device' (default): the residual vectors stay in
    HBM and Levenberg-Marquardt runs on the normal equations of MINPACK's forward-difference Jacobian, reduced on the GPU
    (frank_amd/_levmar.py); 'scipy': every residual vector is copied out and scipy.optimize.least_squares(method='lm')
    drives, exactly as in the reference (at 1e7 visibilities the host-side QR of the 2e7 x 4 Jacobian is then 90 % of the
    time).
    """

    def __init__(self, Rmax, N, inc_pa=None, phase_centre=None, guess=None, verbose=False, device=None, optimizer='device'):
        if optimizer not in _OPTIMIZERS:
            raise ValueError("optimizer must be one of %r, not %r" % (_OPTIMIZERS, optimizer))
        SourceGeometry.__init__(self)  # (inc, PA, dRA, dDec are None until fit() has run)
        self._optimizer = optimizer
        self._N, self._R = N, Rmax
        self._inc_pa, self._phase_centre = inc_pa, phase_centre
        guess = [10., 10., 0., 0.] if guess is None else guess
        if inc_pa is not None:
            guess[0], guess[1] = inc_pa
        if phase_centre is not None:
            guess[2], guess[3] = phase_centre
        self._guess = guess
        self._verbose = verbose
        self._device = device

    def _trial_geometry(self, params):
        inc, pa, dRA, dDec = params
        if self._inc_pa is not None:
            inc, pa = self._inc_pa
        if self._phase_centre is not None:
            dRA, dDec = self._phase_centre
        return FixedGeometry(inc, pa, dRA, dDec)

    @staticmethod
    def _profile_under(geom, DHT, table):
        """FourierBesselFitter(R, N, geom).fit(u, v, vis, w): one binning pass, then GaussianModel without a prior
        (radial_fitters.py:544-582).  Returns (fh_geometry, I)."""
        from frank_amd import _lib
        g, ctx, N = _lib.make_geometry(geom), DHT.context(), DHT.size
        M, j, I = np.empty((N, N)), np.empty(N), np.empty(N)
        H0, qmin, qmax, used_svd = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int(0)
        _lib.check(_lib.lib.fh_ctx_set_scale_height(ctx, None))
        _lib.check(_lib.lib.fh_bin_reset(ctx))
        _lib.check(_lib.lib.fh_bin_visibilities(ctx, ctypes.byref(g), table.handle, 0, table.n))
        _lib.check(_lib.lib.fh_stats_finalize(ctx, ctypes.byref(g), _lib.VIS_MODELS['opt_thick'], 0, _lib.ptr(M), _lib.ptr(j),
                                              ctypes.byref(H0), ctypes.byref(qmin), ctypes.byref(qmax)))
        _lib.check(_lib.lib.fh_gaussian_model(ctx, _lib.ptr(M), _lib.ptr(j), None, _lib.ptr(I), None, None,
                                              ctypes.byref(used_svd)))
        return g, I

    def _report(self, sumsq, n, geom):
        print('\n      FitGeometryFourierBessel: Iteration {}, chi^2={:.8f}, inc={:.3f} PA={:.3f} dRA={:.5f} dDec={:.5f}'
              ''.format(self._counter, 0.5 * sumsq / n, geom.inc, geom.PA, geom.dRA, geom.dDec), end='', flush=True)
        self._counter += 1

    def _residual(self, params, uvdata=None):
        """sqrt(w) (V_model - V), real parts then imaginary parts, of the prior-free fit under the geometry `params`
        (geometry.py:660-694).  uvdata: (DiscreteHankelTransform, _ResidentTable)."""
        from frank_amd import _lib
        DHT, table = uvdata
        geom = self._trial_geometry(params)
        n = table.n
        g, I = self._profile_under(geom, DHT, table)
        out, ss = np.empty(2 * n), ctypes.c_double()
        _lib.check(_lib.lib.fh_vis_residuals(DHT.context(), ctypes.byref(g), _lib.VIS_MODELS['opt_thick'], table.handle, 0, n,
                                             _lib.ptr(I), _lib.ptr(out), ctypes.byref(ss)))
        if self._verbose:
            self._report(ss.value, n, geom)
        return out

    def _fit_on_device(self, DHT, table):
        """Levenberg-Marquardt with the residual vectors in the table's device slots: slot `base` holds r at the current
        point, `spare` the trial point's, four more the forward-difference points'."""
        from frank_amd import _lib
        from frank_amd._levmar import forward_steps, levenberg_marquardt
        free = [k for k in range(4) if (k < 2 and self._inc_pa is None) or (k >= 2 and self._phase_centre is None)]
        x_full = np.array(self._guess, dtype=np.float64)
        slots = {'base': 0, 'spare': 1}
        thick = _lib.VIS_MODELS['opt_thick']

        def full(xf):
            x = x_full.copy()
            x[free] = xf
            return x

        ctx = DHT.context()
        _lib.check(_lib.lib.fh_ctx_set_scale_height(ctx, None))
        H0, qmin, qmax, used_svd = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int(0)

        def evaluate(xf, slot):
            # the whole evaluation on the device: statistics, the prior-free solve and the residual vector never leave it
            geom = self._trial_geometry(full(xf))
            g = _lib.make_geometry(geom)
            ss = ctypes.c_double()
            _lib.check(_lib.lib.fh_bin_reset(ctx))
            _lib.check(_lib.lib.fh_bin_visibilities(ctx, ctypes.byref(g), table.handle, 0, table.n))
            _lib.check(_lib.lib.fh_stats_finalize(ctx, ctypes.byref(g), thick, 0, None, None, ctypes.byref(H0), ctypes.byref(qmin),
                                                  ctypes.byref(qmax)))
            _lib.check(_lib.lib.fh_gaussian_model(ctx, None, None, None, None, None, None, ctypes.byref(used_svd)))
            _lib.check(_lib.lib.fh_vis_residuals_slot(ctx, ctypes.byref(g), thick, table.handle, None, slot, ctypes.byref(ss)))
            if self._verbose:
                self._report(ss.value, table.n, geom)
            return ss.value

        def trial(xf):
            return evaluate(xf, slots['spare'])

        def accept():
            slots['base'], slots['spare'] = slots['spare'], slots['base']

        def normal(xf):
            h = forward_steps(xf)
            cols = (ctypes.c_int * 4)(2, 3, 4, 5)
            for k in range(len(free)):
                xk = np.array(xf, dtype=np.float64)
                xk[k] += h[k]
                evaluate(xk, cols[k])
            A, g = np.empty((len(free), len(free))), np.empty(len(free))
            _lib.check(_lib.lib.fh_residual_normal_equations(DHT.context(), table.handle, slots['base'], len(free), cols,
                                                             _lib.ptr(np.ascontiguousarray(h)), _lib.ptr(A), _lib.ptr(g)))
            return A, g, len(free)
        # (least_squares hands MINPACK maxfev = 100 n (n + 1) with n = 4, whatever is pinned)
        xf, info, _ = levenberg_marquardt(trial, accept, normal, x_full[free], maxfev=2000)
        return full(xf), info in (1, 2, 3, 4)

    def fit(self, u, v, vis, w):
        if self._inc_pa and self._phase_centre:
            logging.info('    You requested a nonparametric fit to determine the geometry, but you provided values for '
                         'inclination, PA, and the phase offset. --> Using your provided values (not fitting for the '
                         'geometry)')
            self._inc, self._PA = self._inc_pa
            self._dRA, self._dDec = self._phase_centre
            return
        from frank_amd.hankel import DiscreteHankelTransform
        logging.info('    Fitting nonparametric form to determine geometry' +
                     (' (your supplied inclination and position angle will be applied at the end of the geometry '
                      'fitting routine)' if self._inc_pa else
                      ' (your supplied phase center will be applied at the end of the geometry fitting routine)'
                      if self._phase_centre else ''))
        DHT = DiscreteHankelTransform(self._R / rad_to_arcsec, self._N, device=self._device)
        table = _ResidentTable(DHT.device, u, v, vis, w)
        self._counter = 0
        try:
            if self._optimizer == 'device':
                best, success = self._fit_on_device(DHT, table)
            else:
                from frank_amd import _lib
                result = _lib.require_scipy("optimizer='scipy' of the geometry fits").least_squares(
                    self._residual, self._guess, kwargs={'uvdata': (DHT, table)}, method='lm')
                best, success = result.x, result.success
        finally:
            table.close()
        if not success:
            raise RuntimeError("FitGeometryFourierBessel failed to converge")
        inc, pa, dRA, dDec = best
        if self._inc_pa:
            inc, pa = self._inc_pa
        else:
            inc, pa = _fix_inc_and_PA_ranges(inc, pa)
        if self._phase_centre:
            dRA, dDec = self._phase_centre
        self._inc, self._PA, self._dRA, self._dDec = inc, pa, dRA, dDec

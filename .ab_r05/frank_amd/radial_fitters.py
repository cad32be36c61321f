"""FourierBesselFitter / FrankFitter / FrankRadialFit -- drop-ins for frank/radial_fitters.py:35-991.

Same constructor signatures, methods, properties, defaults and error behaviour; the arithmetic runs on
the MI355X through libfrank_hip:
  preprocess_visibilities -> fh_map_visibilities (K1 bin_gram)
  fit_preprocessed / _fit -> fh_fit_normal       (K2: the whole power-spectrum loop on the device)
                          -> fh_fit_lognormal    (method='LogNormal': Newton MAP + the same loop, lognormal kernel)
Result objects hold NumPy arrays only and pickle like the reference's (io.py:190).
"""
import abc
import ctypes
from collections import defaultdict
import logging

import numpy as np

from frank_amd import _lib
from frank_amd.constants import rad_to_arcsec
from frank_amd.filter import CriticalFilter
from frank_amd.hankel import DiscreteHankelTransform
from frank_amd.statistical_models import GaussianModel, LogNormalMAPModel, VisibilityMapping


def _forward(owner, attr, doc, scale=None):
    """Read-only property returning getattr(self.<owner>, attr) (times `scale`, e.g. radians -> arcsec)."""
    def get(self):
        val = getattr(getattr(self, owner), attr)
        return val if scale is None else val * scale
    return property(get, doc=doc)


class FrankRadialFit(metaclass=abc.ABCMeta):
    """Base class for results of frank fits (radial_fitters.py:35-219)."""

    def __init__(self, vis_map, info, geometry):
        self._vis_map = vis_map
        self._geometry = geometry
        self._info = info

    def predict(self, u, v, I=None, geometry=None):
        r"""Predict the visibilities in the sky-plane (radial_fitters.py:56-98)."""
        if geometry is None:
            geometry = self._geometry
        if I is None:
            I = self.I
        if geometry is not None:
            # one pass on the device (fh_predict_sky): deproject, H(q) I, scale, re-phase -- at 1e7 baselines the NumPy
            # deprojection and phasor of the lines below are 0.4 s, four times the fit
            return self._vis_map.predict_sky(I, u, v, geometry)
        # no geometry at all (radial_fitters.py:88-90): the baselines are taken as deprojected
        return self._vis_map.predict_visibilities(I, np.hypot(u, v), np.zeros_like(u), geometry=geometry)

    def interpolate_brightness(self, Rpts, I=None):
        """radial_fitters.py:146-176: the brightness profile at the radii Rpts (arcsec) by the Fourier-Bessel series.  I: the
        profile at the collocation points; None takes the MAP (what the reference documents -- its code hands None on)."""
        Rpts = np.array(Rpts)
        if I is None:
            I = self.I
        return self._vis_map.interpolate(I, Rpts, space='Real')

    def predict_deprojected(self, q=None, I=None, geometry=None, block_size=10 ** 5,
                            assume_optically_thick=True):
        r"""Predict the visibilities in the deprojected-plane (radial_fitters.py:100-144)."""
        if geometry is None:
            geometry = self._geometry
        if I is None:
            I = self.I
        if q is None:
            q = self.q
        return self._vis_map.predict_visibilities(I, q, q * 0, geometry=geometry)

    @abc.abstractproperty
    def MAP(self):
        pass

    @property
    def I(self):
        return self.MAP

    r = _forward("_vis_map", "r", "Radius points, unit = arcsec")
    Rmax = _forward("_vis_map", "Rmax", "Maximum radius, unit = arcsec")
    q = _forward("_vis_map", "q", "Frequency points, unit = lambda")
    Qmax = _forward("_vis_map", "Qmax", "Maximum frequency, unit = lambda")
    size = _forward("_vis_map", "size", "Number of points in the reconstruction")
    geometry = property(lambda self: self._geometry, doc="SourceGeometry used for the fit")
    info = property(lambda self: self._info, doc="Hyper-parameters and sizes that reproduce the fit")


class FrankGaussianFit(FrankRadialFit):
    """Result of a frank fit with a Gaussian brightness model (radial_fitters.py:222-322)."""

    def __init__(self, DHT, fit, info={}, geometry=None):
        FrankRadialFit.__init__(self, DHT, info, geometry)
        self._fit = fit

    def draw(self, N):
        return np.random.multivariate_normal(self.mean, self.covariance, N)

    def log_likelihood(self, I=None):
        return self._fit.log_likelihood(I)

    def solve_non_negative(self):
        return self._fit.solve_non_negative()

    mean = _forward("_fit", "mean", "Posterior mean, unit = Jy / sr")
    MAP = _forward("_fit", "mean", "Posterior maximum (= mean for the Gaussian model), unit = Jy / sr")
    covariance = _forward("_fit", "covariance", "Posterior covariance, unit = (Jy / sr)**2")
    power_spectrum = _forward("_fit", "power_spectrum", "Power spectrum coefficients")


class FrankLogNormalFit(FrankRadialFit):
    """Result of a frank fit with a log-normal brightness model (radial_fitters.py:325-403)."""

    def __init__(self, DHT, fit, info={}, geometry=None):
        FrankRadialFit.__init__(self, DHT, info, geometry)
        self._fit = fit

    def log_likelihood(self, I=None):
        return self._fit.log_likelihood(None if I is None else np.log(I))

    @property
    def MAP(self):
        """Posterior maximum, unit = Jy / sr"""
        return np.exp((self._fit.MAP + self._fit.s_0) * self._fit.scale)

    covariance = _forward("_fit", "covariance", "Posterior covariance of s = log I at the MAP")
    power_spectrum = _forward("_fit", "power_spectrum", "Power spectrum coefficients")


class FourierBesselFitter(object):
    """Fourier-Bessel series model for fitting visibilities (radial_fitters.py:405-613).

    Rmax is in arcsec here and radians at the DHT level (:441).
    """

    def __init__(self, Rmax, N, geometry, nu=0, block_data=True, assume_optically_thick=True, scale_height=None,
                 block_size=10 ** 5, verbose=True, device=None, arithmetic='fp64'):
        Rmax /= rad_to_arcsec
        self._geometry = geometry
        # `device` (not in the reference): HIP device of this fitter's GPU work, default $FRANK_AMD_DEVICE or 0
        self._DHT = DiscreteHankelTransform(Rmax, N, nu, device=device)
        if assume_optically_thick:
            if scale_height is not None:
                raise ValueError("Optically thick models must have zero scale-height")
            model = 'opt_thick'
        elif scale_height is not None:
            model = 'debris'
        else:
            model = 'opt_thin'
        self._vis_map = VisibilityMapping(self._DHT, geometry, model, scale_height=scale_height, arithmetic=arithmetic,
                                          block_data=block_data, block_size=block_size, check_qbounds=False,
                                          verbose=verbose)
        self._info = {'Rmax': self._DHT.Rmax * rad_to_arcsec, 'N': self._DHT.size}
        self._verbose = verbose

    def preprocess_visibilities(self, u, v, V, weights=1):
        r"""Prepare the visibilities for fitting (radial_fitters.py:468-498): one bin_gram pass on the GPU."""
        return self._vis_map.map_visibilities(u, v, V, weights)

    def _build_matrices(self, mapping):
        """radial_fitters.py:500-514"""
        self._vis_map.check_hash(mapping['hash'])
        self._M = mapping['M']
        self._j = mapping['j']
        self._H0 = mapping['null_likelihood']

    def fit_method(self):
        """Name of the fit method"""
        return type(self).__name__

    def fit_preprocessed(self, preproc_vis):
        r"""Fit the pre-processed visibilties (radial_fitters.py:520-542)."""
        if self._verbose:
            logging.info('  Fitting pre-processed visibilities for brightness'
                         ' profile using {}'.format(self.fit_method()))
        self._build_matrices(preproc_vis)
        return self._fit()

    def fit(self, u, v, V, weights=1):
        r"""Fit the visibilties (radial_fitters.py:544-572)."""
        if self._verbose:
            logging.info('  Fitting for brightness profile using {}'.format(self.fit_method()))
        self._geometry.fit(u, v, V, weights)
        mapping = self.preprocess_visibilities(u, v, V, weights)
        self._build_matrices(mapping)
        return self._fit()

    def _fit(self):
        """Fit step without a prior (radial_fitters.py:574-582)."""
        fit = GaussianModel(self._DHT, self._M, self._j, noise_likelihood=self._H0)
        self._sol = FrankGaussianFit(self._vis_map, fit, self._info, geometry=self._geometry.clone())
        return self._sol

    r = _forward("_DHT", "r", "Radius points, unit = arcsec", rad_to_arcsec)
    Rmax = _forward("_DHT", "Rmax", "Maximum radius, unit = arcsec", rad_to_arcsec)
    q = _forward("_DHT", "q", "Frequency points, unit = lambda")
    Qmax = _forward("_DHT", "Qmax", "Maximum frequency, unit = lambda")
    size = _forward("_DHT", "size", "Number of points in the reconstruction")
    geometry = property(lambda self: self._geometry, doc="Geometry object")


class FrankFitter(FourierBesselFitter):
    """Gaussian-process fit with the MAP power spectrum (radial_fitters.py:616-991), method 'Normal' or 'LogNormal'.

    Same defaults as the reference: alpha=1.05, p_0=1e-15 ('Normal') / 1e-35 ('LogNormal'), weights_smooth=1e-4,
    tol=1e-3, I_scale=1e5, max_iter=2000, convergence_failure='raise'.

    Three arguments the reference does not have:
      device                 GPU of the fit (default $FRANK_AMD_DEVICE or 0).
      arithmetic             'fp64' (default).  'fp32': single-precision design block and tile products in the binning
                             pass, for tables of at most 2e6 visibilities (RuntimeError beyond: the single-precision
                             Gram loses positive definiteness).  float32 / complex64 input arrays are a different
                             thing: they are STORED in single precision (20 B per visibility) and binned in fp64.
      lognormal_linesearch   method='LogNormal' only.  NOTE: the default, 'linear', is NOT the reference's line-search
                             arithmetic: it forms S^-1 (x + lam p) as S^-1 x + lam S^-1 p instead of multiplying every
                             trial point out (minimizer.py:74-184, statistical_models.py:1088-1113).  Same minimiser,
                             same exit tests, but the Armijo test no longer trips over the round-off of the 1e35-sized
                             entries of S^-1: ~3 x fewer Newton steps and ~10 x less time, a brightness profile within
                             ~1e-5 of its maximum of the reference's -- as far as the reference is from itself after
                             a 1e-15 perturbation of M (tests/golden/lognormal_N300_*.npz).  'reference' reproduces the
                             reference's arithmetic, Newton counters included.
    """

    def __init__(self, Rmax, N, geometry, nu=0, block_data=True, block_size=10 ** 5, alpha=1.05, p_0=None,
                 weights_smooth=1e-4, tol=1e-3, method='Normal', I_scale=1e5, max_iter=2000, check_qbounds=True,
                 store_iteration_diagnostics=False, assume_optically_thick=True, scale_height=None, verbose=True,
                 convergence_failure='raise', device=None, arithmetic='fp64', lognormal_linesearch='linear'):
        if method not in {'Normal', 'LogNormal'}:
            raise ValueError('FrankFitter supports following mehods:\n\t{ "Normal", "LogNormal"}"')
        if lognormal_linesearch not in _lib.LOGNORMAL_LINESEARCH:
            raise ValueError("lognormal_linesearch must be one of %r, not %r" % (_lib.LOGNORMAL_LINESEARCH,
                                                                                  lognormal_linesearch))
        self._method = method
        self._lognormal_linesearch = lognormal_linesearch
        super(FrankFitter, self).__init__(Rmax, N, geometry, nu, block_data, assume_optically_thick, scale_height,
                                          block_size, verbose, device=device, arithmetic=arithmetic)
        # Reinstate the bounds check: FourierBesselFitter does not check bounds (radial_fitters.py:706-707)
        self._vis_map.check_qbounds = check_qbounds
        if p_0 is None:
            p_0 = 1e-15 if method == 'Normal' else 1e-35
        self._s_scale = np.log(I_scale)
        self._filter = CriticalFilter(self._DHT, alpha, p_0, weights_smooth, tol)
        self._max_iter = max_iter
        self._store_iteration_diagnostics = store_iteration_diagnostics
        self._info.update({'alpha': alpha, 'wsmooth': weights_smooth, 'p0': p_0, 'method': method})
        if convergence_failure not in {'raise', 'warn', 'ignore'}:
            raise ValueError("convergence_failure must be one of 'raise',"
                             f"'warn', or 'ignore', nor {convergence_failure}")
        self._convergence_failure = convergence_failure
        self._hyper = (float(alpha), float(p_0), float(weights_smooth), float(tol))

    def fit_method(self):
        """Name of the fit method"""
        return '{}: {} method'.format(type(self).__name__, self._method)

    def _fit(self):
        """The power-spectrum iteration (radial_fitters.py:737-832), run on the device by fh_fit_normal /
        fh_fit_lognormal; the convergence policy (:787-815) is applied to the returned `count`."""
        N = self.size
        alpha, p_0, wsmooth, tol = self._hyper
        lognormal = self._method == 'LogNormal'
        x, p = np.empty(N), np.empty(N)
        niter = ctypes.c_int(0)
        dp = dm = None
        if self._store_iteration_diagnostics:
            self._iteration_diagnostics = defaultdict(list)
            dp = np.zeros((self._max_iter + 1, N))
            dm = np.zeros((self._max_iter + 1, N))
        M, j = _lib.f8(self._M), _lib.f8(self._j)
        if lognormal:
            Dinv = np.empty((N, N))
            stats = (ctypes.c_int64 * 9)()
            _lib.set_lognormal_linesearch(self._DHT.context(), self._lognormal_linesearch)
            rc = _lib.lib.fh_fit_lognormal(self._DHT.context(), _lib.ptr(M), _lib.ptr(j), alpha, p_0, wsmooth, tol,
                                           int(self._max_iter), float(np.exp(self._s_scale)), _lib.ptr(x), _lib.ptr(p),
                                           ctypes.byref(niter), _lib.ptr(Dinv), stats, _lib.ptr(dp), _lib.ptr(dm))
        else:
            rc = _lib.lib.fh_fit_normal(self._DHT.context(), _lib.ptr(M), _lib.ptr(j), alpha, p_0, wsmooth, tol,
                                        int(self._max_iter), _lib.ptr(x), _lib.ptr(p), ctypes.byref(niter),
                                        _lib.ptr(dp), _lib.ptr(dm))
        if rc == _lib.FH_ERR_BAD_P:
            from frank_amd.statistical_models import _BAD_P_MSG
            raise ValueError(_BAD_P_MSG)
        if rc == _lib.FH_ERR_NOT_SPD:
            # a Cholesky inside the device loop failed (for method='LogNormal': in one of the two Normal seed solves,
            # radial_fitters.py:744-752, or of the Hessian at a MAP, statistical_models.py:1150-1158): the reference
            # carries on through the SVD pseudo-inverse (statistical_models.py:747-755); so does the loop below, one
            # posterior at a time.  (Logged: it is ~10 x slower than the fused loop, and a fit that takes this route for
            # no numerical reason -- it once did, at the sizes where a library inverse was wrong -- should be noticed.)
            logging.info('    A Cholesky factorisation failed inside the device loop: continuing one posterior at a '
                         'time through the SVD route (as the reference does; slower)')
            return self._fit_one_posterior_at_a_time()
        _lib.check(rc)
        count = niter.value

        if self._store_iteration_diagnostics:
            self._iteration_diagnostics['power_spectrum'] = [dp[i].copy() for i in range(count)]
            self._iteration_diagnostics['MAP'] = [dm[i].copy() for i in range(count)]

        self._check_convergence_policy(count)

        if self._store_iteration_diagnostics:
            self._iteration_diagnostics['num_iterations'] = count

        if lognormal:
            fit = LogNormalMAPModel._from_solution(self._DHT, self._M, self._j, p, x, Dinv, self._s_scale,
                                                   noise_likelihood=self._H0, stats=tuple(stats))
            self._sol = FrankLogNormalFit(self._vis_map, fit, self._info, geometry=self._geometry.clone())
        else:
            fit = GaussianModel._from_solution(self._DHT, self._M, self._j, p, x, noise_likelihood=self._H0)
            self._sol = FrankGaussianFit(self._vis_map, fit, self._info, geometry=self._geometry.clone())
        self._ps = p
        self._ps_cov = None
        return self._sol

    def _fit_one_posterior_at_a_time(self):
        """radial_fitters.py:743-832 step by step, either method: every posterior is a GaussianModel / LogNormalMAPModel
        (device Cholesky, device SVD pseudo-inverse when that fails), every update a
        CriticalFilter.update_power_spectrum.  The fused device loops stop at the first failed Cholesky; this is where
        such a fit continues."""
        N = self.size
        lognormal = self._method == 'LogNormal'
        if self._store_iteration_diagnostics:
            self._iteration_diagnostics = defaultdict(list)
        pI = np.ones(N)
        fit = self._perform_fit(pI, guess=np.ones_like(pI), fit_method='Normal')
        pI = np.max(self._DHT.transform(fit.MAP) ** 2)
        pI = pI * (self.q / self.q[0]) ** -2
        fit = self._perform_fit(pI, fit_method='Normal')
        if lognormal:  # radial_fitters.py:756-763
            s = np.log(np.maximum(fit.MAP, 1e-3 * fit.MAP.max()))
            s -= self._s_scale
            pI = np.max(self._DHT.transform(s) ** 2)
            pI = pI * (self.q / self.q[0]) ** -4
            fit = self._perform_fit(pI, guess=s)
        count = 0
        pi_old = 0
        while (not self._filter.check_convergence(pI, pi_old)) and count <= self._max_iter:
            pi_old = pI.copy()
            pI = self._filter.update_power_spectrum(fit)
            fit = self._perform_fit(pI, guess=fit.MAP)
            if self._store_iteration_diagnostics:
                self._iteration_diagnostics['power_spectrum'].append(pI)
                self._iteration_diagnostics['MAP'].append(fit.MAP)
            count += 1
        self._check_convergence_policy(count)
        if self._store_iteration_diagnostics:
            self._iteration_diagnostics['num_iterations'] = count
        Sol = FrankLogNormalFit if lognormal else FrankGaussianFit
        self._sol = Sol(self._vis_map, fit, self._info, geometry=self._geometry.clone())
        self._ps = pI
        self._ps_cov = None
        return self._sol

    def _check_convergence_policy(self, count):
        """radial_fitters.py:787-815: success iff count < max_iter; otherwise raise / warn / ignore."""
        if count < self._max_iter:
            if self._verbose:
                logging.info('    Converged after {} power-spectrum iterations'.format(count))
            return
        msg = ('Convergence not met within {} iterations. Increase max_iter or alpha (convergence is slow for '
               'alpha close to 1)'.format(self._max_iter))
        if self._convergence_failure == 'raise':
            raise RuntimeError(msg + ", or set convergence_failure to 'warn' / 'ignore' to keep the last iterate.")
        if self._convergence_failure == 'warn':
            if logging.getLogger().isEnabledFor(logging.INFO):
                logging.info(msg)
            else:
                print(msg)

    def _perform_fit(self, p, guess=None, fit_method=None):
        """Posterior for a given p (radial_fitters.py:858-890)."""
        if fit_method is None:
            fit_method = self._method
        if fit_method == 'Normal':
            return GaussianModel(self._DHT, self._M, self._j, p, guess=guess, noise_likelihood=self._H0)
        if fit_method == 'LogNormal':
            return LogNormalMAPModel(self._DHT, self._M, self._j, p, guess=guess, s0=self._s_scale,
                                     noise_likelihood=self._H0, linesearch=self._lognormal_linesearch)
        raise ValueError('fit_method must be one of the following:\n\t{"Normal", "LogNormal"}')

    def draw_powerspectrum(self, Ndraw=1):
        """radial_fitters.py:834-856"""
        log_p = np.random.multivariate_normal(np.log(self._ps), self.MAP_spectrum_covariance, Ndraw)
        return np.exp(log_p)

    def log_prior(self, p=None):
        """radial_fitters.py:892-919"""
        if p is None:
            p = self._ps
        return self._filter.log_prior(p)

    def log_likelihood(self, sol=None):
        r"""radial_fitters.py:922-949"""
        if sol is None:
            sol = self.MAP_solution
        return self.log_prior(sol.power_spectrum) + sol.log_likelihood()

    def log_evidence_laplace(self):
        r"""radial_fitters.py:951-967"""
        Sigma_inv = self._filter.covariance_MAP(self._sol, ret_inv=True)
        sign, logdet = np.linalg.slogdet(Sigma_inv / (2 * np.pi))
        return self.log_likelihood() - 0.5 * logdet

    MAP_solution = property(lambda self: self._sol, doc="Reconstruction for the maximum a posteriori power spectrum")
    MAP_spectrum = property(lambda self: self._ps, doc="Maximum a posteriori power spectrum")

    @property
    def MAP_spectrum_covariance(self):
        """Covariance matrix of the maximum a posteriori power spectrum"""
        if self._ps_cov is None:
            self._ps_cov = self._filter.covariance_MAP(self._sol)
        return self._ps_cov

    iteration_diagnostics = property(lambda self: self._iteration_diagnostics,
                                     doc="dict: power spectrum and posterior mean of every iteration, num_iterations")

"""VisibilityMapping, GaussianModel and LogNormalMAPModel -- drop-ins for frank/statistical_models.py:29-1295.

Every array computation here is a call into libfrank_hip (HIP kernels, rocBLAS, rocSOLVER); the Python
keeps the reference's signatures, attribute names, error behaviour and plain-NumPy results.
"""
import ctypes
import logging

import numpy as np

from frank_amd import _lib
from frank_amd.constants import rad_to_arcsec, deg_to_rad


class VisibilityMapping:
    r"""Builds the mapping between the visibility and image planes (statistical_models.py:29-568).

    Same constructor as the reference.  `block_data` / `block_size` are accepted for compatibility: the
    GPU kernel streams the visibility axis itself, so they do not change the result's meaning (the
    reference's result depends on them only through summation order, ~1e-15 relative).
    """

    def __init__(self, DHT, geometry, vis_model='opt_thick', scale_height=None, block_data=True,
                 block_size=10 ** 5, check_qbounds=True, verbose=True, arithmetic='fp64'):
        _vis_models = ['opt_thick', 'opt_thin', 'debris']
        if arithmetic not in ('fp64', 'fp32'):
            raise ValueError("arithmetic must be 'fp64' or 'fp32'")
        if arithmetic == 'fp32' and vis_model == 'debris':
            raise ValueError("single-precision binning is not built for the debris model")
        # (not in the reference) 'fp32': the table is stored in single precision and the Bessel design block and the
        # tile products of the Gram run in single precision on the matrix pipe, with fp64 argument reduction and fp64
        # block accumulation (fh_ctx_set_arithmetic); M, j and the fit stay fp64
        self._arithmetic = arithmetic
        if vis_model not in _vis_models:
            raise ValueError(f"vis_model must be one of {_vis_models}")  # statistical_models.py:71-73
        if vis_model == 'debris' and scale_height is None:
            raise ValueError('You requested a model with a non-zero scale height'
                             ' but did not specify H(R) (scale_height=None)')
        self._vis_model = vis_model
        self.check_qbounds = check_qbounds
        self._verbose = verbose
        self._chunking = block_data
        self._chunk_size = block_size
        self._DHT = DHT
        self._geometry = geometry
        self._scale_height = None
        if vis_model == 'debris':  # statistical_models.py:96-102
            self._scale_height = scale_height(self.r)
            self._H2 = 0.5 * (2 * np.pi * self._scale_height / rad_to_arcsec) ** 2
        if self._verbose:
            if vis_model == 'opt_thick':
                logging.info('  Assuming an optically thick model (the default): '
                             'Scaling the total flux to account for the source inclination')
            else:
                logging.info('  Assuming an optically thin model: *Not* scaling the '
                             'total flux to account for the source inclination')

    def map_visibilities(self, u, v, V, weights, frequencies=None, geometry=None):
        r"""M = H^T w H, j = H^T w Re(V), H0 (statistical_models.py:109-237) via the bin_gram kernel.

        Returns the reference's dict: keys 'mult_freq', 'channels', 'M', 'j', 'null_likelihood', 'hash'.
        As in the reference the correction always uses the geometry given at construction (:165); the
        `geometry` argument only goes into the hash.
        """
        if geometry is None:
            geometry = self._geometry
        if frequencies is not None:
            return self._map_channels(u, v, V, weights, frequencies, geometry)
        if self._verbose:
            logging.info('    Building visibility matrices M and j')
        V = np.asarray(V)
        f32 = _lib.all_float32(u, v, V, weights) or self._arithmetic == 'fp32'
        N = self.size
        M, j = np.empty((N, N)), np.empty(N)
        H0, qmin, qmax = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        g = _lib.make_geometry(self._geometry)
        ctx = self._DHT.context()
        # geometrically thick model: the kernel needs H2[k] to scale every row by exp(-kz^2 H2[k]) (:494-496)
        _lib.check(_lib.lib.fh_ctx_set_scale_height(
            ctx, _lib.ptr(_lib.f8(self._H2)) if self._vis_model == 'debris' else None))
        _lib.check(_lib.lib.fh_ctx_set_arithmetic(ctx, 1 if self._arithmetic == 'fp32' else 0))
        model = _lib.VIS_MODELS[self._vis_model]
        conv = _lib.f4 if f32 else _lib.f8
        u, v = conv(u), conv(v)
        # a complex128 array goes to the device as it is (re, im interleaved) and is split there: the two strided host copies
        # into separate columns were 30 ms of a 45 ms call at 1e7 visibilities
        as_pairs = (not f32) and V.dtype == np.complex128 and V.ndim == 1 and V.flags.c_contiguous
        Vre = None if as_pairs else conv(V.real)
        Vim = None if as_pairs else (conv(V.imag) if np.iscomplexobj(V) else None)
        w = conv(np.atleast_1d(weights))
        n = u.size
        if v.size != n or V.size != n or w.size not in (1, n):
            raise ValueError("u, v, V (and weights) must have matching lengths")
        if f32:
            # a table handed over in single precision (float32 u, v, weights, complex64 / float32 V) is stored and
            # streamed as fp32, 20 B per visibility, and widened in the pre-pass: the arithmetic is the fp64 path, as the
            # reference's is whatever dtype it gets (NumPy promotes in geometry.py:69-79)
            vis = ctypes.c_void_p()
            _lib.check(_lib.lib.fh_vis_upload_f32(self._DHT.device, _lib.fptr(u), _lib.fptr(v), _lib.fptr(Vre), _lib.fptr(Vim),
                                                  _lib.fptr(w), w.size, n, ctypes.byref(vis)))
            try:
                _lib.check(_lib.lib.fh_bin_reset(ctx))
                _lib.check(_lib.lib.fh_bin_visibilities(ctx, ctypes.byref(g), vis, 0, n))
                _lib.check(_lib.lib.fh_stats_finalize(ctx, ctypes.byref(g), model, 0, _lib.ptr(M), _lib.ptr(j),
                                                      ctypes.byref(H0), ctypes.byref(qmin), ctypes.byref(qmax)))
            finally:
                _lib.lib.fh_vis_destroy(vis)
        elif as_pairs:
            rc = _lib.lib.fh_map_visibilities_c128(
                ctx, ctypes.byref(g), model, 1 if self.check_qbounds else 0, _lib.ptr(u), _lib.ptr(v),
                V.view(np.float64).ctypes.data_as(ctypes.POINTER(ctypes.c_double)), _lib.ptr(w), w.size, n, _lib.ptr(M), _lib.ptr(j),
                ctypes.byref(H0), ctypes.byref(qmin), ctypes.byref(qmax))
            if rc != _lib.FH_ERR_QRANGE:
                _lib.check(rc)
        else:
            rc = _lib.lib.fh_map_visibilities(
                ctx, ctypes.byref(g), model, 1 if self.check_qbounds else 0, _lib.ptr(u), _lib.ptr(v),
                _lib.ptr(Vre), _lib.ptr(Vim), _lib.ptr(w), w.size, n, _lib.ptr(M), _lib.ptr(j), ctypes.byref(H0),
                ctypes.byref(qmin), ctypes.byref(qmax))
            if rc != _lib.FH_ERR_QRANGE:  # (that one: stopped before the binning; the reference's message is raised below)
                _lib.check(rc)
        self._check_uv_range(qmin.value, qmax.value)
        return {
            'mult_freq': False,
            'channels': None,
            'M': M,
            'j': j,
            'null_likelihood': H0.value,
            'hash': [False, self._DHT, geometry, self._vis_model, self._scale_height],
        }

    def _map_channels(self, u, v, V, weights, frequencies, geometry):
        """The multi-frequency form of map_visibilities (statistical_models.py:175-237): one (M, j) per distinct value of
        `frequencies`, one null likelihood for the lot.  The table goes to the device once; a channel is one binning pass over
        it with the rows of the other channels given multiplicity zero (fh_vis_set_multiplicity: the bootstrap's mechanism)."""
        if self._verbose:
            logging.info('    Building visibility matrices M and j')
        V = np.asarray(V)
        f32 = _lib.all_float32(u, v, V, weights)
        conv, ptr, upload = ((_lib.f4, _lib.fptr, _lib.lib.fh_vis_upload_f32) if f32 else
                             (_lib.f8, _lib.ptr, _lib.lib.fh_vis_upload))
        u, v = conv(u), conv(v)
        Vre, Vim = conv(V.real), (conv(V.imag) if np.iscomplexobj(V) else None)
        w = conv(np.atleast_1d(weights))
        frequencies = np.asarray(frequencies)
        n = u.size
        if v.size != n or Vre.size != n or w.size not in (1, n) or frequencies.size != n:
            raise ValueError("u, v, V, frequencies (and weights) must have matching lengths")
        channels = np.unique(frequencies)
        N = self.size
        Ms, js = np.zeros((len(channels), N, N)), np.zeros((len(channels), N))
        g, ctx, model = _lib.make_geometry(self._geometry), self._DHT.context(), _lib.VIS_MODELS[self._vis_model]
        _lib.check(_lib.lib.fh_ctx_set_scale_height(
            ctx, _lib.ptr(_lib.f8(self._H2)) if self._vis_model == 'debris' else None))
        _lib.check(_lib.lib.fh_ctx_set_arithmetic(ctx, 1 if self._arithmetic == 'fp32' else 0))
        table = ctypes.c_void_p()
        _lib.check(upload(self._DHT.device, ptr(u), ptr(v), ptr(Vre), ptr(Vim), ptr(w), w.size, n, ctypes.byref(table)))
        H0_all, q_lo, q_hi = 0.0, np.inf, 0.0
        try:
            for i, f in enumerate(channels):
                member = np.ascontiguousarray(frequencies == f, dtype=np.int32)
                H0, qmin, qmax = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
                _lib.check(_lib.lib.fh_vis_set_multiplicity(table, member.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
                _lib.check(_lib.lib.fh_bin_reset(ctx))
                _lib.check(_lib.lib.fh_bin_visibilities(ctx, ctypes.byref(g), table, 0, n))
                _lib.check(_lib.lib.fh_stats_finalize(ctx, ctypes.byref(g), model, 0, _lib.ptr(Ms[i]), _lib.ptr(js[i]),
                                                      ctypes.byref(H0), ctypes.byref(qmin), ctypes.byref(qmax)))
                H0_all += H0.value
                q_lo, q_hi = min(q_lo, qmin.value), max(q_hi, qmax.value)
        finally:
            _lib.lib.fh_vis_destroy(table)
        self._check_uv_range(q_lo, q_hi)
        return {
            'mult_freq': True,
            'channels': channels,
            'M': Ms,
            'j': js,
            'null_likelihood': H0_all,
            'hash': [True, self._DHT, geometry, self._vis_model, self._scale_height],
        }

    def check_hash(self, hash, multi_freq=False, geometry=None):
        """statistical_models.py:239-276"""
        if geometry is None:
            geometry = self._geometry
        passed = (
            multi_freq == hash[0] and
            self._DHT.Rmax == hash[1].Rmax and
            self._DHT.size == hash[1].size and
            self._DHT.order == hash[1].order and
            geometry.inc == hash[2].inc and
            geometry.PA == hash[2].PA and
            geometry.dRA == hash[2].dRA and
            geometry.dDec == hash[2].dDec and
            self._vis_model == hash[3]
        )
        if not passed:
            return False
        if self._scale_height is None:
            return hash[4] is None
        if hash[4] is None:
            return False
        return np.all(self._scale_height == hash[4])

    def _scale(self, geometry=None):
        if self._vis_model == 'opt_thick':  # statistical_models.py:486-490
            if geometry is None:
                geometry = self._geometry
            return np.cos(geometry.inc * deg_to_rad)
        return 1.0

    def interpolate(self, f, r, space='Real'):
        """statistical_models.py:435-481: f (at the collocation points) interpolated to the points r -- arcsec in 'Real' space,
        lambda in 'Fourier' space --, in chunks when the mapping was built with block_data."""
        if space == 'Real':
            r = r / rad_to_arcsec
        r = np.array(r)
        shape = r.shape
        r = r.reshape(-1)
        Ni = int(self._chunk_size / len(r) + 1) if self._chunking else len(r)
        out, end = [], 0
        while end < len(r):
            start, end = end, end + Ni
            out.append(self._DHT.interpolate(f, r[start:end], space))
        return np.concatenate(out).reshape(*shape)

    def predict_visibilities(self, I, q, k=None, geometry=None):
        r"""V(q) = H(q) I on the GPU (statistical_models.py:279-329)."""
        q = _lib.f8(np.atleast_1d(q)).reshape(-1)
        I = _lib.f8(I)
        if I.size != self.size:
            raise ValueError("I must have one value per collocation point")
        if self._vis_model == 'debris':
            # H(q) from the GPU, the per-(visibility, column) factor exp(-k^2 H2) and the product on the host
            return np.dot(self._get_mapping_coefficients(q, np.asarray(k, dtype=np.float64).reshape(-1)), I)
        V = np.empty(q.size)
        _lib.check(_lib.lib.fh_predict_visibilities(self._DHT.context(), _lib.ptr(q), q.size, _lib.ptr(I),
                                                    float(self._scale(geometry)), _lib.ptr(V)))
        return V

    def predict_sky(self, I, u, v, geometry):
        """Model visibilities at sky-plane baselines (u, v) under `geometry`: FrankRadialFit.predict's deproject,
        predict_visibilities and undo_correction (radial_fitters.py:85-98) as one device pass (the debris model's
        exp(-kz^2 H2[k]) per column included)."""
        shape = np.shape(u)
        u, v = _lib.f8(np.ravel(u)), _lib.f8(np.ravel(v))
        I = _lib.f8(I)
        if I.size != self.size:
            raise ValueError("I must have one value per collocation point")
        if u.size != v.size:
            raise ValueError("u and v must have matching lengths")
        Vre, Vim = np.empty(u.size), np.empty(u.size)
        g = _lib.make_geometry(geometry)
        ctx = self._DHT.context()
        _lib.check(_lib.lib.fh_ctx_set_scale_height(
            ctx, _lib.ptr(_lib.f8(self._H2)) if self._vis_model == 'debris' else None))
        _lib.check(_lib.lib.fh_predict_sky(ctx, ctypes.byref(g), _lib.VIS_MODELS[self._vis_model], _lib.ptr(u), _lib.ptr(v),
                                           u.size, _lib.ptr(I), _lib.ptr(Vre), _lib.ptr(Vim)))
        return (Vre + 1j * Vim).reshape(shape)

    def invert_visibilities(self, V, R, geometry=None):
        r"""statistical_models.py:331-384 (backward coefficients on the GPU, 1/scale applied)."""
        R = np.atleast_1d(R)
        H = self._DHT._device_coefficients(R / rad_to_arcsec, 'backward', 1.0 / self._scale(geometry))
        return np.dot(H, V)[R < self.Rmax]

    def transform(self, f, q=None, direction='forward'):
        """statistical_models.py:386-412"""
        if direction == 'backward' and q is not None:
            q = q / rad_to_arcsec
        return self._DHT.transform(f, q, direction)

    def DHT_coefficients(self, direction='forward'):
        """statistical_models.py:414-433"""
        return self._DHT.coefficients(direction=direction)

    def _get_mapping_coefficients(self, qs, ks, geometry=None, inverse=False):
        """H(q) with the model's scale (statistical_models.py:483-509), built on the GPU."""
        if self._vis_model == 'debris':
            scale = np.exp(-np.outer(np.asarray(ks) * np.asarray(ks), self._H2))
            if inverse:
                return self._DHT._device_coefficients(np.asarray(qs) / rad_to_arcsec, 'backward', 1.0) * \
                    np.atleast_1d(1 / scale).reshape(1, -1)
            return self._DHT._device_coefficients(qs, 'forward', 1.0) * scale
        scale = self._scale(geometry)
        if inverse:
            return self._DHT._device_coefficients(np.asarray(qs) / rad_to_arcsec, 'backward', 1.0 / scale)
        return self._DHT._device_coefficients(qs, 'forward', scale)

    def _check_uv_range(self, uvmin, uvmax=None):
        """statistical_models.py:512-535 (takes the min / max the kernel reduced, or an array of baselines)."""
        if uvmax is None:
            uv = np.asarray(uvmin)
            uvmin, uvmax = uv.min(), uv.max()
        if self.check_qbounds:
            if self.q[0] < uvmin:
                logging.warning(r"WARNING: First collocation point, q[0] = {:.3e} \lambda,"
                                " is at a baseline shorter than the"
                                " shortest deprojected baseline in the dataset,"
                                r" min(uv) = {:.3e} \lambda. For q[0] << min(uv),"
                                " the fit's total flux may be biased"
                                " low.".format(self.q[0], uvmin))
            if self.q[-1] < uvmax:
                raise ValueError(r"ERROR: Last collocation point, {:.3e} \lambda, is at"
                                 " a shorter baseline than the longest deprojected"
                                 r" baseline in the dataset, {:.3e} \lambda. Please"
                                 " increase N in FrankMultFrequencyFitter (this is"
                                 " `hyperparameters: n` if you're using a parameter"
                                 " file). Or if you'd like to fit to shorter maximum baseline,"
                                 " cut the (u, v) distribution before fitting"
                                 " (`modify_data: baseline_range` in the"
                                 " parameter file).".format(self.q[-1], uvmax))

    @property
    def r(self):
        """Radius points, unit = arcsec"""
        return self._DHT.r * rad_to_arcsec

    @property
    def Rmax(self):
        """Maximum radius, unit = arcsec"""
        return self._DHT.Rmax * rad_to_arcsec

    @property
    def q(self):
        r"""Frequency points, unit = :math:`\lambda`"""
        return self._DHT.q

    @property
    def Qmax(self):
        r"""Maximum frequency, unit = :math:`\lambda`"""
        return self._DHT.Qmax

    @property
    def size(self):
        """Number of points in reconstruction"""
        return self._DHT.size

    @property
    def scale_height(self):
        return self._scale_height


_BAD_P_MSG = ("Bad value in power spectrum. The power"
              " spectrum must be positive and not contain"
              " any NaN values. This is likely due to"
              " your UVtable (incorrect units or weights), "
              " or the deprojection being applied (incorrect"
              " geometry and/or phase center). Else you may"
              " want to adjust `rout` (ensure it is larger than"
              " the source) or `n` (up to ~300).")


class GaussianModel:
    r"""Posterior of the Bayesian linear regression, D = (M + S(p)^-1)^-1, mu = D j
    (statistical_models.py:571-904), single field / single channel, solved on the GPU
    (rocBLAS dgemm + rocSOLVER potrf/potrs through fh_gaussian_model).
    """

    def __init__(self, DHT, M, j, p=None, scale=None, guess=None, Nfields=None, noise_likelihood=0):
        self._DHT = DHT
        M = np.asarray(M, dtype=np.float64)
        j = np.asarray(j, dtype=np.float64)
        if M.ndim == 3 and M.shape[0] == 1:
            M = M[0]
        if j.ndim == 2 and j.shape[0] == 1:
            j = j[0]
        if M.ndim != 2 or j.ndim != 1 or scale is not None or (Nfields not in (None, 1)):
            raise NotImplementedError("multi-channel / multi-field GaussianModel (statistical_models.py:655-726) "
                                      "is outside the hot path built so far")
        self._Nfields = 1
        if p is not None:
            p = np.asarray(p, dtype=np.float64).reshape(-1)
            if np.any(p <= 0) or np.any(np.isnan(p)):  # statistical_models.py:688-698
                print(p)
                raise ValueError(_BAD_P_MSG)
        self._p = p
        self._M = np.ascontiguousarray(M)
        self._j = np.ascontiguousarray(j)
        self._like_noise = noise_likelihood
        self._Sinv = None
        self._cov = None
        self._Dchol = None
        self._used_svd = False
        self._mu = None
        self._fit()

    @classmethod
    def _from_solution(cls, DHT, M, j, p, mu, noise_likelihood=0):
        """Wrap a posterior the device loop already solved (no recomputation; factor rebuilt on demand)."""
        self = cls.__new__(cls)
        self._DHT = DHT
        self._Nfields = 1
        self._p = np.asarray(p, dtype=np.float64)
        self._M = np.ascontiguousarray(M, dtype=np.float64)
        self._j = np.ascontiguousarray(j, dtype=np.float64)
        self._like_noise = noise_likelihood
        self._Sinv = None
        self._cov = None
        self._Dchol = None
        self._used_svd = False
        self._mu = np.asarray(mu, dtype=np.float64)
        return self

    def _fit(self, want_sinv=False):
        """statistical_models.py:732-760"""
        N = self.size
        mu, chol = np.empty(N), np.empty((N, N))
        Sinv = np.empty((N, N)) if want_sinv else None
        used_svd = ctypes.c_int(0)
        p = None if self._p is None else _lib.f8(self._p)
        _lib.check(_lib.lib.fh_gaussian_model(self._DHT.context(), _lib.ptr(self._M), _lib.ptr(self._j), _lib.ptr(p),
                                              _lib.ptr(mu), _lib.ptr(chol), _lib.ptr(Sinv), ctypes.byref(used_svd)))
        self._used_svd = bool(used_svd.value)
        if self._mu is None or not self._used_svd:
            self._mu = mu
        self._Dchol = None if self._used_svd else chol
        if want_sinv:
            self._Sinv = Sinv if self._p is not None else None
        self._cov = None

    def _ensure_factor(self):
        if self._Dchol is None and not self._used_svd:
            mu_keep = self._mu
            self._fit()
            self._mu = mu_keep

    def Dsolve(self, b):
        r"""Compute D . b by solving D^-1 x = b (statistical_models.py:762-781)."""
        self._ensure_factor()
        b = np.asarray(b, dtype=np.float64)
        if self._Dchol is None:
            # Cholesky failed for this Dinv: the reference's SVD route (:779-781), rocSOLVER gesvd on the device
            Dinv = self._M + (self._sinv() if self._p is not None else 0)
            return _svd_solve(self._DHT, Dinv, b)
        shape = b.shape
        B = np.array(b.reshape(self.size, -1), dtype=np.float64, order='C')  # copy: the solve is in place
        _lib.check(_lib.lib.fh_cho_solve(self._DHT.context(), _lib.ptr(self._Dchol), _lib.ptr(B), B.shape[1]))
        return B.reshape(shape)

    def _sinv(self):
        if self._Sinv is None and self._p is not None:
            mu_keep = self._mu
            self._fit(want_sinv=True)
            self._mu = mu_keep
        return self._Sinv

    def draw(self, N):
        """Compute N draws from the posterior (statistical_models.py:783-788; host RNG)."""
        return np.random.multivariate_normal(self.mean.reshape(-1), self.covariance, N)

    def log_likelihood(self, I=None):
        r"""statistical_models.py:790-856 (host slogdet; the solves run on the GPU)."""
        Sinv = self._sinv()
        if I is None:
            like = 0.5 * np.sum(self._j * self._mu)
            if Sinv is not None:
                Q = self.Dsolve(Sinv)
                like += 0.5 * np.linalg.slogdet(Q)[1]
        else:
            Dinv = self._M + (Sinv if Sinv is not None else 0)
            like = np.sum(self._j * I) - 0.5 * np.dot(I, np.dot(Dinv, I))
            if Sinv is not None:
                like += 0.5 * np.linalg.slogdet(2 * np.pi * Sinv)[1]
        return like + self._like_noise

    def solve_non_negative(self):
        """statistical_models.py:858-866 (SciPy NNLS on the host; off the hot path)."""
        from frank_amd import _lib
        nnls = _lib.require_scipy("solve_non_negative (scipy.optimize.nnls)").nnls
        Sinv = self._sinv()
        Dinv = self._M + (Sinv if Sinv is not None else 0)
        return nnls(Dinv, self._j, maxiter=100 * len(self._j))[0]

    @property
    def mean(self):
        """Posterior mean, unit = Jy / sr"""
        return self._mu

    @property
    def MAP(self):
        """Posterior maximum, unit = Jy / sr"""
        return self.mean

    @property
    def covariance(self):
        """Posterior covariance, unit = (Jy / sr)**2"""
        if self._cov is None:
            self._cov = self.Dsolve(np.eye(self.size))
        return self._cov

    @property
    def s_0(self):
        return 0

    @property
    def power_spectrum(self):
        """Power spectrum coefficients"""
        return self._p

    @property
    def num_fields(self):
        return self._Nfields

    @property
    def size(self):
        """Number of points in reconstruction"""
        return self._DHT.size



def _svd_solve(DHT, Dinv, b):
    """np.dot(V.T, np.multiply(np.dot(U.T, b), s1)) with U, s, V = svd(Dinv), s1 = where(s > 0, 1/s, 0), as the
    reference writes it (statistical_models.py:779-781): rocSOLVER gesvd + rocBLAS on the device.  For a vector that is
    the pseudo-inverse solve; for an N x N `b` NumPy broadcasts s1 over the last axis (column c times s1[c]) and the
    reference's power-spectrum iteration runs on exactly that, so it is reproduced (fh_svd_solve_as_reference)."""
    b = np.asarray(b, dtype=np.float64)
    N = DHT.size
    B = np.array(b.reshape(N, -1), dtype=np.float64, order='C')
    if b.ndim > 1 and B.shape[1] != N:
        raise ValueError("operands could not be broadcast together with shapes (%d,%d) (%d,) " % (N, B.shape[1], N))
    entry = _lib.lib.fh_svd_solve if b.ndim == 1 else _lib.lib.fh_svd_solve_as_reference
    _lib.check(entry(DHT.context(), _lib.ptr(_lib.f8(Dinv)), _lib.ptr(B), B.shape[1]))
    return B.reshape(b.shape)


class LogNormalMAPModel:
    r"""Maximum a posteriori log-brightness, P(s|q,V,p,s0) ~ G(H exp(s+s0) - V, M) P(s|p)
    (statistical_models.py:907-1295), for one field and one frequency with scale = 1 -- what FrankFitter builds
    (radial_fitters.py:885-887).  The Newton minimisation (minimizer.py) runs in the lognormal kernel through
    fh_lognormal_model; `Dsolve` applies the inverse of the Hessian at the MAP.
    """

    def __init__(self, DHT, M, j, p=None, scale=None, s0=None, guess=None, Nfields=None, full_hessian=1,
                 noise_likelihood=0, linesearch='linear'):
        # linesearch (not in the reference): 'linear' forms S^-1 (x + lam p) from S^-1 x and S^-1 p along a line search,
        # 'reference' multiplies every trial point out as minimizer.py does (include/frank_hip.h)
        if linesearch not in _lib.LOGNORMAL_LINESEARCH:
            raise ValueError("linesearch must be one of %r, not %r" % (_lib.LOGNORMAL_LINESEARCH, linesearch))
        self._linesearch = linesearch
        self._DHT = DHT
        M = np.asarray(M, dtype=np.float64)
        j = np.asarray(j, dtype=np.float64)
        if M.ndim == 3 and M.shape[0] == 1:
            M = M[0]
        if j.ndim == 2 and j.shape[0] == 1:
            j = j[0]
        if (M.ndim != 2 or j.ndim != 1 or Nfields not in (None, 1)
                or (scale is not None and np.any(np.asarray(scale) != 1))):
            raise NotImplementedError("multi-frequency / multi-field LogNormalMAPModel (statistical_models.py:"
                                      "1017-1047) is unreachable from FrankFitter and not built")
        if full_hessian != 1:
            raise NotImplementedError("full_hessian != 1 (statistical_models.py:1112-1120) is not built; FrankFitter "
                                      "always uses the full Hessian")
        if s0 is None or guess is None:
            # the reference fails on both as well (s0.reshape / guess.reshape on None, :1058, :1124)
            raise ValueError("LogNormalMAPModel needs s0 and guess")
        s0 = np.atleast_1d(np.asarray(s0, dtype=np.float64))
        if len(s0) != 1:
            raise ValueError("Signal zero-point (s0) must have the same "
                             "length as the number of fields or length 1")
        self._Nfields = 1
        self._full_hess = full_hessian
        self._scale = np.ones([1, 1], dtype='f8')
        self._s0 = s0.reshape(1, 1)
        N = DHT.size
        if p is not None:
            p = np.asarray(p, dtype=np.float64).reshape(-1, N)
            if np.any(p <= 0) or np.any(np.isnan(p)):  # statistical_models.py:1049-1057
                raise ValueError(_BAD_P_MSG)
        self._p = p
        self._M = np.ascontiguousarray(M).reshape(1, N, N)
        self._j = np.ascontiguousarray(j).reshape(1, N)
        self._like_noise = noise_likelihood
        self._Sinv_cache = None
        self._cov = None
        self._Dchol = None
        self._chol_failed = False
        self._fit(np.asarray(guess, dtype=np.float64).reshape(N))

    @classmethod
    def _from_solution(cls, DHT, M, j, p, s_map, Dinv, s0, noise_likelihood=0, stats=None):
        """Wrap a MAP the device loop already found (fh_fit_lognormal)."""
        self = cls.__new__(cls)
        N = DHT.size
        self._DHT = DHT
        self._linesearch = 'linear'
        self._Nfields = 1
        self._full_hess = 1
        self._scale = np.ones([1, 1], dtype='f8')
        self._s0 = np.full((1, 1), float(s0))
        self._p = np.asarray(p, dtype=np.float64).reshape(1, N)
        self._M = np.ascontiguousarray(M, dtype=np.float64).reshape(1, N, N)
        self._j = np.ascontiguousarray(j, dtype=np.float64).reshape(1, N)
        self._like_noise = noise_likelihood
        self._Sinv_cache = None
        self._cov = None
        self._Dchol = None
        self._chol_failed = False
        self._s_MAP = np.asarray(s_map, dtype=np.float64).reshape(1, N)
        self._Dinv = np.asarray(Dinv, dtype=np.float64)
        self._newton_stats = stats
        return self

    def _fit(self, guess):
        """statistical_models.py:1064-1160"""
        N = self.size
        s_map, Dinv = np.empty(N), np.empty((N, N))
        stats = (ctypes.c_int64 * 9)()
        # no prior (p=None): S^-1 = 0 (:1063) is p -> infinity
        p = np.full(N, np.inf) if self._p is None else _lib.f8(self._p[0])
        _lib.set_lognormal_linesearch(self._DHT.context(), self._linesearch)
        _lib.check(_lib.lib.fh_lognormal_model(self._DHT.context(), _lib.ptr(_lib.f8(self._M[0])),
                                               _lib.ptr(_lib.f8(self._j[0])), _lib.ptr(p), _lib.ptr(_lib.f8(guess)),
                                               float(self._s0[0, 0]), _lib.ptr(s_map), _lib.ptr(Dinv), stats))
        self._s_MAP = s_map.reshape(1, N)
        self._Dinv = Dinv
        self._newton_stats = tuple(stats)
        self._cov = None

    @property
    def _Sinv(self):
        """Prior precision Y^T diag(1/p) Y, shape (1, N, N) (statistical_models.py:1060-1063)."""
        if self._Sinv_cache is None:
            N = self.size
            if self._p is None:
                self._Sinv_cache = np.zeros([1, N, N], dtype='f8')
            else:
                Ykm = self._DHT.coefficients()
                self._Sinv_cache = np.einsum('ji,lj,jk->lik', Ykm, 1 / self._p, Ykm)
        return self._Sinv_cache

    def _ensure_factor(self):
        if self._Dchol is None and not self._chol_failed:
            N = self.size
            mu, chol = np.empty(N), np.empty((N, N))
            used_svd = ctypes.c_int(0)
            sym = np.ascontiguousarray(0.5 * (self._Dinv + self._Dinv.T))
            _lib.check(_lib.lib.fh_gaussian_model(self._DHT.context(), _lib.ptr(sym), _lib.ptr(np.zeros(N)), None,
                                                  _lib.ptr(mu), _lib.ptr(chol), None, ctypes.byref(used_svd)))
            if used_svd.value:
                self._chol_failed = True
            else:
                self._Dchol = chol

    def Dsolve(self, b):
        r"""Compute D . b by solving D^-1 x = b, D^-1 = hess(s_MAP) (statistical_models.py:1162-1182)."""
        self._ensure_factor()
        b = np.asarray(b, dtype=np.float64)
        if self._Dchol is None:
            # the Hessian at the MAP is not positive definite: the reference's SVD route (:1150-1158), on the device
            return _svd_solve(self._DHT, self._Dinv, b)
        shape = b.shape
        B = np.array(b.reshape(self.size, -1), dtype=np.float64, order='C')
        _lib.check(_lib.lib.fh_cho_solve(self._DHT.context(), _lib.ptr(self._Dchol), _lib.ptr(B), B.shape[1]))
        return B.reshape(shape)

    def draw(self, N):
        """Compute N draws from the (approximate) posterior (statistical_models.py:1184-1189; host RNG)."""
        return np.random.multivariate_normal(self.MAP.reshape(-1), self.covariance, N)

    def log_likelihood(self, s=None):
        r"""statistical_models.py:1191-1241.  The reference's own implementation raises for every input
        (`np.einsum('i,j->ij', self._scale, s)` with a 2-D `_scale`), so there is no behaviour to match."""
        raise NotImplementedError("LogNormalMAPModel.log_likelihood raises in the reference as well "
                                  "(statistical_models.py:1232)")

    def solve_non_negative(self):
        """The solution is always non-negative; provided for convenience (statistical_models.py:1243-1246)."""
        return self.MAP

    @property
    def MAP(self):
        """Posterior maximum of s = log(I) - s0"""
        return self._s_MAP.reshape(self.size)

    @property
    def covariance(self):
        """Posterior covariance at the MAP"""
        if self._cov is None:
            self._cov = self.Dsolve(np.eye(self.size))
        return self._cov

    @property
    def power_spectrum(self):
        """Power spectrum coefficients"""
        return None if self._p is None else self._p.reshape(self.size)

    @property
    def scale(self):
        return self._scale[:, 0]

    @property
    def s_0(self):
        return self._s0[0]

    @property
    def num_fields(self):
        return self._Nfields

    @property
    def size(self):
        """Number of points in reconstruction"""
        return self._DHT.size

"""UVDataBinner and estimate_weights -- drop-ins for frank/utilities.py:180-400 and :515-631.

The row-wise work (bin index of every baseline, weighted sums per bin, squared deviations from the bin mean) runs in
the uvbin kernels of libfrank_hip (LDS histograms, one streaming pass per statistic); what is left on the host is
O(nbins): masking, and the neighbour logic of estimate_weights.
"""
import ctypes
import logging

import numpy as np

from frank_amd import _lib
from frank_amd.hankel import default_device

_i32p = ctypes.POINTER(ctypes.c_int32)
_i64p = ctypes.POINTER(ctypes.c_int64)


class UVDataBinner(object):
    r"""Average uv-data into bins of equal size: the weighted mean of the visibilities in each bin
    (utilities.py:180-400).  Same constructor and properties as the reference; empty bins are masked.

    Parameters
    ----------
    uv : array, unit = :math:`\lambda` -- baselines of the data to bin
    V : array, unit = Jy -- observed visibility; if complex both components are binned
    weights : array, unit = Jy^-2 -- weights on the visibility points
    bin_width : float, unit = :math:`\lambda`
    """

    def __init__(self, uv, V, weights, bin_width):
        uv = _lib.f8(uv)
        V = np.asarray(V)
        self._complex = np.iscomplexobj(V)
        Vre = _lib.f8(V.real)
        Vim = _lib.f8(V.imag) if self._complex else None
        w = _lib.f8(np.broadcast_to(weights, uv.shape))
        if Vre.shape != uv.shape:
            raise ValueError("uv and V must have the same length")
        self._handle = ctypes.c_void_p()
        _lib.check(_lib.lib.fh_uvbin_create(default_device(), _lib.ptr(uv), _lib.ptr(Vre), _lib.ptr(Vim), _lib.ptr(w), uv.size,
                                            float(bin_width), ctypes.byref(self._handle)))
        nbins = self._nbins = _lib.lib.fh_uvbin_nbins(self._handle)
        self._bins = np.arange(nbins + 1, dtype='float64') * bin_width
        self._norm = 1 / bin_width
        b_uv, b_re, b_im, b_w = (np.empty(nbins) for _ in range(4))
        e_re, e_im = np.empty(nbins), np.empty(nbins)
        b_n = np.empty(nbins, dtype=np.int64)
        _lib.check(_lib.lib.fh_uvbin_get(self._handle, _lib.ptr(b_uv), _lib.ptr(b_re), _lib.ptr(b_im), _lib.ptr(b_w),
                                         b_n.ctypes.data_as(_i64p), _lib.ptr(e_re), _lib.ptr(e_im)))
        mask = (b_n == 0)
        b_V = (b_re + 1j * b_im) if self._complex else b_re
        err = (e_re + 1j * e_im) if self._complex else e_re
        err[mask] = np.nan
        self._uv = np.ma.masked_where(mask, b_uv)
        self._V = np.ma.masked_where(mask, b_V)
        self._w = np.ma.masked_where(mask, b_w)
        self._count = np.ma.masked_where(mask, b_n)
        self._uv_left = np.ma.masked_where(mask, self._bins[:-1])
        self._uv_right = np.ma.masked_where(mask, self._bins[1:])
        self._Verr = np.ma.masked_where(mask, err)

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            _lib.lib.fh_uvbin_destroy(h)
            self._handle = None

    def determine_uv_bin(self, uv):
        r"""Bin that each of the given uv points belongs to; -1 if the bin does not exist (utilities.py:271-298)."""
        uv = _lib.f8(uv)
        idx = np.empty(uv.shape, dtype=np.int32)
        rc = _lib.lib.fh_uvbin_determine(self._handle, _lib.ptr(uv), uv.size, idx.ctypes.data_as(_i32p))
        if rc == _lib.FH_ERR_INVALID:
            raise IndexError(_lib.last_error())
        _lib.check(rc)
        return idx

    def bin_quantities(self, uv, w, *quantities, bin_counts=False):
        r"""Bin the given quantities according to the uv points and weights (utilities.py:300-366)."""
        uv, w = _lib.f8(uv), _lib.f8(w)
        nbins = self._nbins
        results = []
        counts = np.zeros(nbins, dtype=np.int64) if bin_counts else None
        for k, qty in enumerate(quantities):
            qty = np.asarray(qty)
            cplx = np.iscomplexobj(qty)
            qre = _lib.f8(qty.real)
            qim = _lib.f8(qty.imag) if cplx else None
            ore = np.empty(nbins)
            oim = np.empty(nbins) if cplx else None
            want = counts if (bin_counts and k == 0) else None
            rc = _lib.lib.fh_uvbin_quantities(self._handle, _lib.ptr(uv), _lib.ptr(w), _lib.ptr(qre), _lib.ptr(qim),
                                              uv.size, _lib.ptr(ore), _lib.ptr(oim),
                                              None if want is None else want.ctypes.data_as(_i64p))
            if rc == _lib.FH_ERR_INVALID:
                raise IndexError(_lib.last_error())
            _lib.check(rc)
            results.append((ore + 1j * oim).astype(qty.dtype) if cplx else ore.astype(qty.dtype, copy=False))
        if bin_counts:
            return results + [counts]
        if len(results) == 1:
            return results[0]
        return results

    def __len__(self):
        return len(self._uv)

    uv = property(lambda self: self._uv, doc=r"Binned uv points, unit = :math:`\lambda`")
    V = property(lambda self: self._V, doc="Binned visibility, unit = Jy")
    weights = property(lambda self: self._w, doc="Binned weights, unit = Jy^-2")
    error = property(lambda self: self._Verr, doc="Uncertainty on the binned visibilities, unit = Jy")
    bin_counts = property(lambda self: self._count, doc="Number of points in each bin")
    bin_edges = property(lambda self: [self._uv_left, self._uv_right], doc="Edges of the histogram bins")


def estimate_weights(u, v=None, V=None, nbins=300, log=True, use_median=False, verbose=True):
    r"""Estimate the weights from the variance of the binned visibilities (utilities.py:515-631); same call forms:
    `estimate_weights(u, v, V)`, `estimate_weights(u, V)`, `estimate_weights(u, V=V)`.
    """
    if verbose:
        logging.info('  Estimating visibility weights')
    if V is None:
        if v is None:
            raise ValueError("The visibilities, V, must be supplied")
        V, q = v, np.abs(u)
    elif v is not None:
        q = np.hypot(u, v)
    else:
        q = np.abs(u)
    V = np.asarray(V)
    if log:
        q = np.log(q)
        q -= q.min()
    bin_width = (q.max() - q.min()) / nbins
    uvBin = UVDataBinner(q, V, np.ones_like(q), bin_width)
    if uvBin.bin_counts.max() == 1:
        raise ValueError("No bin contains more than one uv point, can't"
                         " estimate the variance. Use fewer bins.")
    # as in the reference, `np.iscomplex(V.dtype)` is False for every dtype: the real-part variance is used (:589-592)
    var = uvBin.error.real ** 2 * uvBin.bin_counts
    if use_median:
        if verbose:
            logging.info('    Setting all weights as median binned visibility variance')
        return np.full(len(u), 1 / np.ma.median(var[uvBin.bin_counts > 1]))
    if verbose:
        logging.info('    Setting weights according to baseline-dependent binned visibility variance')
    single = np.argwhere(uvBin.bin_counts == 1).reshape(-1)
    if len(single) > 0:  # bins with one point: mean of the two adjacent bins that have a variance (:606-616)
        good = np.argwhere(uvBin.bin_counts > 1).reshape(-1)
        loc = np.searchsorted(good, single, side='right')
        below = good[np.maximum(loc - 1, 0)]
        above = good[np.minimum(loc, len(good) - 1)]
        var[single] = 0.5 * (var[below] + var[above])
    bin_id = uvBin.determine_uv_bin(q)
    assert np.all(bin_id != -1), "Error in binning"
    return 1 / var[bin_id]


# ---- callers of the transform: mock data and direct transforms (utilities.py:634-666, 923-1146) ----------------------
def draw_bootstrap_sample(u, v, vis, weights):
    """One bootstrap resample of the data set: len(u) rows drawn with replacement (utilities.py:634-666; the global NumPy
    generator, one randint call).  `frank_amd.bootstrap.bootstrap_fits` draws the same indices but keeps the table on the
    device and hands the kernel row multiplicities instead of gathered copies."""
    pick = np.random.randint(low=0, high=len(u), size=len(u))
    return u[pick], v[pick], vis[pick], weights[pick]


def add_vis_noise(vis, weights, seed=None):
    """Visibilities plus Gaussian noise of standard deviation weights**-0.5, independently on the real and (for complex
    input) imaginary parts (utilities.py:923-959).  The draws come from the global NumPy generator in the reference's
    order -- one standard_normal((1 or 2,) + vis.shape) call after the optional np.random.seed(seed) -- so a seeded call
    returns the reference's numbers."""
    if seed is not None:
        np.random.seed(seed)
    vis = np.array(vis)
    parts = 2 if np.iscomplexobj(vis) else 1
    draws = np.random.standard_normal((parts,) + vis.shape)
    draws *= weights ** -0.5
    noisy = vis + draws[0]
    if parts == 2:
        noisy += 1j * draws[1]
    return noisy


def get_collocation_points(Rmax=2.0, N=500, direction='forward'):
    """Collocation points of the transform for (Rmax [arcsec], N): radii in arcsec ('forward') or spatial frequencies in
    lambda ('backward') (utilities.py:1041-1074)."""
    if direction not in ['forward', 'backward']:
        raise AttributeError("direction must be one of ['forward', 'backward']")
    from frank_amd.constants import rad_to_arcsec
    from frank_amd.hankel import DiscreteHankelTransform
    r_pts, q_pts = DiscreteHankelTransform.get_collocation_points(Rmax=Rmax / rad_to_arcsec, N=N, nu=0)
    return r_pts * rad_to_arcsec if direction == 'forward' else q_pts


def generic_dht(x, f, Rmax=2.0, N=500, direction='forward', grid=None, inc=0.0):
    """Visibilities of a brightness profile f(x [arcsec]) ('forward') or the profile of visibilities f(x [lambda])
    ('backward') by the discrete Hankel transform, sampled at `grid` (default: the collocation points) and scaled by
    cos(inc) as an optically thick disc (utilities.py:1077-1146).  f is interpolated linearly onto the collocation points;
    the transform itself runs on the GPU (VisibilityMapping.predict_visibilities / invert_visibilities).
    Returns (grid, transform)."""
    if direction not in ['forward', 'backward']:
        raise AttributeError("direction must be one of ['forward', 'backward']")
    from frank_amd.constants import rad_to_arcsec
    from frank_amd.geometry import FixedGeometry
    from frank_amd.hankel import DiscreteHankelTransform
    from frank_amd.statistical_models import VisibilityMapping
    face_on = FixedGeometry(inc, 0, 0, 0)
    VM = VisibilityMapping(DiscreteHankelTransform(Rmax=Rmax / rad_to_arcsec, N=N, nu=0), face_on)
    if direction == 'forward':
        grid = VM.q if grid is None else grid
        return grid, VM.predict_visibilities(np.interp(VM.r, x, f), grid, geometry=face_on)
    grid = VM.r if grid is None else grid
    return grid, VM.invert_visibilities(np.interp(VM.q, x, f), grid, geometry=face_on)


def make_mock_data(r, I, Rmax, u, v, projection=None, geometry=None, N=500, add_noise=False, weights=None, seed=None):
    """Mock visibilities of a profile I(r [arcsec]) at the baselines (u, v) (utilities.py:962-1038): optionally
    deproject / reproject the baselines with `geometry` (whose inclination then scales the flux), transform with
    generic_dht, optionally add noise for the given weights.  Returns (baselines, vis)."""
    allowed = [None, 'deproject', 'reproject']
    if projection not in allowed:
        raise AttributeError(f"projection is '{projection}'; must be one of {allowed}.")
    from frank_amd.geometry import FixedGeometry
    if projection is None:
        if geometry is not None:
            raise AttributeError("projection is None; must be one of ['deproject', 'reproject'] to perform projection.")
        geometry = FixedGeometry(0, 0, 0, 0)
    else:
        if geometry is None:
            raise AttributeError(f"geometry must be supplied to perform {projection}.")
        u, v = geometry.deproject(u, v) if projection == 'deproject' else geometry.reproject(u, v)
    baselines = np.hypot(u, v)
    _, vis = generic_dht(r, I, Rmax, N, grid=baselines, inc=geometry.inc)
    if add_noise:
        vis = add_vis_noise(vis, weights, seed)
    return baselines, vis


# ---- data preparation either side of the path: units and cuts (utilities.py:31-177, 403-512; host arithmetic, no device work) ----
def arcsec_baseline(x):
    """A radial scale [arcsec] as the baseline [lambda] that resolves it, or the other way round: the map is its own
    inverse, 1 / (x arcsec in radians) (utilities.py:31-50)."""
    return 1 / (x / 60 / 60 * np.pi / 180)


def radius_convert(x, dist, conversion='arcsec_au'):
    """Radii between [arcsec] and [au] for a source at `dist` [pc] (utilities.py:53-83)."""
    if conversion == 'arcsec_au':
        return x * dist
    if conversion == 'au_arcsec':
        return x / dist
    raise AttributeError("conversion must be one of {}".format(['arcsec_au', 'au_arcsec']))


_JY_CONVERSIONS = ['beam_sterad', 'beam_arcsec2', 'arcsec2_beam', 'arcsec2_sterad', 'sterad_beam', 'sterad_arcsec2']


def jy_convert(x, conversion, bmaj=None, bmin=None):
    """Brightness between [Jy / beam], [Jy / arcsec^2] and [Jy / sterad]; `conversion` names source_target, e.g.
    'beam_sterad'; conversions through the beam need its FWHMs bmaj, bmin [arcsec] (utilities.py:86-138)."""
    from frank_amd.constants import sterad_to_arcsec
    have_beam = bmaj is not None and bmin is not None
    if not have_beam and conversion in ['beam_sterad', 'beam_arcsec2', 'arcsec2_beam', 'sterad_beam']:
        raise ValueError('bmaj and bmin must be specified to perform the conversion {}'.format(conversion))
    if conversion not in _JY_CONVERSIONS:
        raise AttributeError("conversion must be one of {}".format(_JY_CONVERSIONS))
    beam = np.pi * bmaj * bmin / (4 * np.log(2)) if have_beam else None  # solid angle of a Gaussian beam [arcsec^2]
    if conversion == 'beam_arcsec2':
        return x / beam
    if conversion == 'arcsec2_beam':
        return x * beam
    if conversion == 'arcsec2_sterad':
        return x * sterad_to_arcsec
    if conversion == 'sterad_arcsec2':
        return x / sterad_to_arcsec
    if conversion == 'beam_sterad':
        return x / beam * sterad_to_arcsec
    return x * beam / sterad_to_arcsec  # 'sterad_beam'


def get_fit_stat_uncer(fit, return_linear=True):
    """1-sigma statistical uncertainty of a fitted profile from the diagonal of its covariance (a lower bound: the sparse
    (u, v) sampling adds a systematic part); for a LogNormal fit the variance of log I is turned into that of I unless
    return_linear is False (utilities.py:141-177)."""
    if 'method' not in fit._info.keys():
        raise AttributeError("'fit' object lacks '_info.method' key. Should be one of ['linear', 'log']")
    variance = np.diag(fit.covariance)
    if fit._info["method"] == "LogNormal" and return_linear == True:  # noqa: E712  (the reference's comparison)
        variance = (np.exp(variance) - 1) * np.exp(2 * np.log(fit.I))
    return np.sqrt(variance)


def check_uv(u, v, min_q=1e3, max_q=1e8):
    """Warn when the shortest baseline is below min_q: the table is then probably in metres, not wavelengths
    (utilities.py:403-428; max_q is accepted and, as in the reference, not looked at)."""
    q = np.hypot(u, v)
    if min(q) < min_q:
        logging.warning("WARNING: "
                        f"Minimum baseline {min(q):.1e} < expected minimum {min_q:.1e} [lambda]. "
                        "'u' and 'v' distances must be in units of [lambda], but it looks like they're in [m].")


def normalize_uv(u, v, wle):
    """(u, v) in metres -> wavelengths: divided by the observing wavelength(s) `wle` [m], one value or one per row
    (utilities.py:431-460)."""
    logging.info('  Normalizing u and v coordinates by provided observing wavelength of {} m'.format(wle))
    wle = np.atleast_1d(wle).astype('f8')
    if len(wle) != 1 and len(wle) != len(u):
        raise ValueError("len(wle) = {}. It should be equal to len(u) = {} (or 1 if all wavelengths are the same)".format(
            len(wle), len(u)))
    return u / wle, v / wle


def cut_data_by_baseline(u, v, vis, weights, cut_range, geometry=None):
    """Rows whose baseline -- deprojected with `geometry` if given -- lies inside cut_range = [min, max] lambda, both ends
    included (utilities.py:463-512).  Returns (u, v, vis, weights) of those rows."""
    logging.info('  Cutting data outside of the minimum and maximum baselines of {} and {} klambda'.format(
        cut_range[0] / 1e3, cut_range[1] / 1e3))
    q = np.hypot(*(geometry.deproject(u, v) if geometry is not None else (u, v)))
    keep = (q >= cut_range[0]) & (q <= cut_range[1])
    return u[keep], v[keep], vis[keep], weights[keep]

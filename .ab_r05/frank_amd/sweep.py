"""Hyper-parameter sweeps over one mapping: the batched form of frank/fit.py:534-548 (`run_multiple_fits`).

The reference re-runs the whole fit (including the visibility mapping) for every (alpha, w_smooth) point
although M and j do not depend on them.  Here the mapping is done once (FrankFitter.preprocess_visibilities)
and all points are iterated concurrently, one workgroup (one compute unit) per point: fh_fit_normal_batched
(fit_loop kernel) for method='Normal', fh_fit_lognormal_batched (lognormal kernel) for method='LogNormal'."""
import ctypes

import numpy as np

from frank_amd import _lib
from frank_amd.radial_fitters import FrankFitter, FrankGaussianFit, FrankLogNormalFit
from frank_amd.statistical_models import GaussianModel, LogNormalMAPModel, _BAD_P_MSG


def split_grid(npoints, ndevices):
    """Contiguous, near-equal index ranges [(first, count), ...] of `npoints` sweep points for `ndevices` devices
    (broadcast (M, j), split the fits evenly, no further communication)."""
    from frank_amd.distributed import shard_range
    return [shard_range(npoints, d, ndevices) for d in range(ndevices)]


def sweep_fits(fitter, preproc_vis, alphas, weights_smooth, p_0=None, tol=1e-3, max_iter=2000, devices=None):
    """Fit `preproc_vis` (from `fitter.preprocess_visibilities`) for every (alpha[i], weights_smooth[i]).

    Returns (sols, niters): FrankGaussianFit (FrankLogNormalFit for a method='LogNormal' fitter) objects as
    FrankFitter.fit_preprocessed would return for a fitter constructed with those hyper-parameters, and the iteration
    counts (`count`; >= max_iter means not converged).

    devices : list of HIP device indices, optional.  The grid is split evenly over them (split_grid), M and j go to
        every device once (the only transfer), and each device iterates its points concurrently with the others: no
        communication between devices.  Default: the fitter's own device.  The results do not depend on the split
        (every point is one workgroup running the same arithmetic wherever it is placed).
    """
    if not isinstance(fitter, FrankFitter):
        raise TypeError("fitter must be a frank_amd FrankFitter")
    alphas = _lib.f8(np.atleast_1d(alphas))
    ws = _lib.f8(np.atleast_1d(weights_smooth))
    if alphas.shape != ws.shape:
        raise ValueError("alphas and weights_smooth must have the same length")
    B, N = alphas.size, fitter.size
    lognormal = fitter._method == 'LogNormal'
    p0 = _lib.f8(np.full(B, (1e-35 if lognormal else 1e-15) if p_0 is None else p_0))
    fitter._build_matrices(preproc_vis)
    M, j = _lib.f8(fitter._M), _lib.f8(fitter._j)
    devices = [fitter._DHT.device] if not devices else [int(d) for d in devices]
    if lognormal:
        return _sweep_lognormal(fitter, M, j, alphas, ws, p0, tol, max_iter, devices)
    mu, p = np.empty((B, N)), np.empty((B, N))
    niter = np.zeros(B, dtype=np.intc)
    status = np.zeros(B, dtype=np.intc)
    ip = ctypes.POINTER(ctypes.c_int)

    def run(dev, first, count):
        if count == 0:
            return
        sl = slice(first, first + count)
        _lib.check(_lib.lib.fh_fit_normal_batched(
            fitter._DHT.context(dev), _lib.ptr(M), _lib.ptr(j), count, _lib.ptr(alphas[sl]), _lib.ptr(p0[sl]),
            _lib.ptr(ws[sl]), float(tol), int(max_iter), _lib.ptr(mu[sl]), _lib.ptr(p[sl]),
            niter[sl].ctypes.data_as(ip), status[sl].ctypes.data_as(ip)))
    _on_devices(run, devices, B)
    sols = []
    for b in range(B):
        if status[b] == _lib.FH_ERR_BAD_P:
            raise ValueError(_BAD_P_MSG)
        if status[b] == _lib.FH_ERR_NOT_SPD:
            # a Cholesky of this point's loop failed: continue it the way the reference does, through the SVD route
            sols.append(_refit_through_svd_route(fitter, float(alphas[b]), float(p0[b]), float(ws[b]), tol, max_iter, niter, b))
            continue
        if status[b] != _lib.FH_OK:
            raise RuntimeError("fit %d of the sweep failed (status %d)" % (b, status[b]))
        fit = GaussianModel._from_solution(fitter._DHT, fitter._M, fitter._j, p[b].copy(), mu[b].copy(),
                                           noise_likelihood=fitter._H0)
        info = dict(fitter._info, alpha=float(alphas[b]), wsmooth=float(ws[b]), p0=float(p0[b]))
        sols.append(FrankGaussianFit(fitter._vis_map, fit, info, geometry=fitter._geometry.clone()))
    return sols, [int(n) for n in niter]


def _on_devices(run, devices, npoints):
    """run(device, first, count) for every slice of split_grid.  Slices of different devices run concurrently (one host
    thread per device; ctypes drops the GIL for the duration of a call, and a context's work is confined to its own
    device and stream); slices that name the same device share its context and run one after the other."""
    parts = split_grid(npoints, len(devices))
    by_dev = {}
    for d, part in zip(devices, parts):
        by_dev.setdefault(d, []).append(part)

    def run_all(d):
        for first, count in by_dev[d]:
            run(d, first, count)
    if len(by_dev) == 1:
        run_all(devices[0])
        return
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=len(by_dev)) as pool:
        futs = [pool.submit(run_all, d) for d in by_dev]
        for fu in futs:
            fu.result()


def _refit_through_svd_route(fitter, alpha, p_0, wsmooth, tol, max_iter, niter, b):
    """One point of a sweep whose device loop hit a failed Cholesky: FrankFitter._fit_one_posterior_at_a_time with this
    point's hyper-parameters (statistical_models.py:747-755 semantics)."""
    import copy
    from frank_amd.filter import CriticalFilter
    sub = copy.copy(fitter)
    sub._filter = CriticalFilter(fitter._DHT, alpha, p_0, wsmooth, tol)
    sub._max_iter = int(max_iter)
    sub._info = dict(fitter._info, alpha=alpha, wsmooth=wsmooth, p0=p_0)
    sub._store_iteration_diagnostics = True
    sub._convergence_failure = 'ignore'  # the sweep reports iteration counts; its caller applies the policy
    sol = sub._fit_one_posterior_at_a_time()
    niter[b] = sub._iteration_diagnostics['num_iterations']
    return sol


def _sweep_lognormal(fitter, M, j, alphas, ws, p0, tol, max_iter, devices):
    B, N = alphas.size, fitter.size
    s_map, p = np.empty((B, N)), np.empty((B, N))
    niter = np.zeros(B, dtype=np.intc)
    status = np.zeros(B, dtype=np.intc)
    stats = np.zeros(9 * B, dtype=np.int64)
    ip, lp = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int64)

    def run(dev, first, count):
        if count == 0:
            return
        sl = slice(first, first + count)
        _lib.set_lognormal_linesearch(fitter._DHT.context(dev), fitter._lognormal_linesearch)
        _lib.check(_lib.lib.fh_fit_lognormal_batched(
            fitter._DHT.context(dev), _lib.ptr(M), _lib.ptr(j), count, _lib.ptr(alphas[sl]), _lib.ptr(p0[sl]),
            _lib.ptr(ws[sl]), float(tol), int(max_iter), float(np.exp(fitter._s_scale)), _lib.ptr(s_map[sl]),
            _lib.ptr(p[sl]), niter[sl].ctypes.data_as(ip), status[sl].ctypes.data_as(ip),
            stats[9 * first:9 * (first + count)].ctypes.data_as(lp)))
    _on_devices(run, devices, B)
    sols = []
    for b in range(B):
        if status[b] == _lib.FH_ERR_BAD_P:
            raise ValueError(_BAD_P_MSG)
        if status[b] != _lib.FH_OK:
            raise RuntimeError("fit %d of the sweep failed (status %d)" % (b, status[b]))
        # the Hessian at the MAP is rebuilt on demand (covariance / Dsolve) by a one-step MAP solve from s_map
        fit = _LazyLogNormal._from_map(fitter._DHT, fitter._M, fitter._j, p[b].copy(), s_map[b].copy(),
                                       fitter._s_scale, fitter._H0, tuple(stats[9 * b:9 * b + 9]))
        info = dict(fitter._info, alpha=float(alphas[b]), wsmooth=float(ws[b]), p0=float(p0[b]))
        sols.append(FrankLogNormalFit(fitter._vis_map, fit, info, geometry=fitter._geometry.clone()))
    return sols, [int(n) for n in niter]


class _LazyLogNormal(LogNormalMAPModel):
    """A sweep result: MAP and power spectrum from the batched kernel; `_Dinv` (the Hessian at the MAP, needed only by
    covariance / Dsolve / update_power_spectrum) is obtained on first use by re-solving from the MAP itself."""

    @classmethod
    def _from_map(cls, DHT, M, j, p, s_map, s0, noise_likelihood, stats):
        self = cls._from_solution(DHT, M, j, p, s_map, np.zeros((0, 0)), s0, noise_likelihood, stats)
        self._lazy = True
        return self

    def __getattribute__(self, name):
        if name == '_Dinv' and object.__getattribute__(self, '__dict__').get('_lazy'):
            self._lazy = False
            keep_s, keep_stats = self._s_MAP.copy(), self._newton_stats
            self._fit(keep_s.reshape(-1))  # converges at once: starts at the MAP
            self._s_MAP, self._newton_stats = keep_s, keep_stats
        return object.__getattribute__(self, name)


def sweep_evidence(fitter, preproc_vis, sols, alphas, weights_smooth, p_0=None, covariance=False, device=None):
    """Rank the points of a sweep: for every solution of `sweep_fits` (same order as alphas / weights_smooth) the posterior's
    marginal likelihood, the log prior of its power spectrum and the Laplace evidence -- FrankFitter.log_likelihood /
    log_prior / log_evidence_laplace (radial_fitters.py:892-967, filter.py:184-263) for ALL points in a few batched device
    calls (fh_sweep_evidence) instead of dense O(N^3) host algebra per point.

    Returns a dict of arrays of length len(sols): 'log_likelihood' (= log_prior + the solution's marginal likelihood, what
    FrankFitter.log_likelihood() returns), 'sol_log_likelihood', 'log_prior', 'log_evidence', and with covariance=True
    'spectrum_covariance_diag' (len(sols) x N: the diagonal of MAP_spectrum_covariance)."""
    alphas = _lib.f8(np.atleast_1d(alphas))
    ws = _lib.f8(np.atleast_1d(weights_smooth))
    B, N = len(sols), fitter.size
    if alphas.size != B or ws.size != B:
        raise ValueError("alphas and weights_smooth must match the solutions")
    if fitter._method == 'LogNormal':
        raise NotImplementedError("sweep_evidence: the Laplace evidence of the reference is defined for method='Normal' "
                                  "(filter.py:184-227 takes the Gaussian posterior's covariance)")
    p0 = _lib.f8(np.full(B, 1e-15 if p_0 is None else p_0))
    fitter._build_matrices(preproc_vis)
    M, j = _lib.f8(fitter._M), _lib.f8(fitter._j)
    p = _lib.f8(np.array([s.power_spectrum for s in sols]).reshape(B, N))
    mu = _lib.f8(np.array([s.I for s in sols]).reshape(B, N))
    sll, lp, lev = np.empty(B), np.empty(B), np.empty(B)
    cov = np.empty((B, N)) if covariance else None
    ctx = fitter._DHT.context(fitter._DHT.device if device is None else device)
    _lib.check(_lib.lib.fh_sweep_evidence(ctx, _lib.ptr(M), _lib.ptr(j), float(fitter._H0), B, _lib.ptr(p), _lib.ptr(mu),
                                          _lib.ptr(alphas), _lib.ptr(p0), _lib.ptr(ws), _lib.ptr(sll), _lib.ptr(lp), _lib.ptr(lev),
                                          _lib.ptr(cov) if covariance else None))
    out = {'log_likelihood': lp + sll, 'sol_log_likelihood': sll, 'log_prior': lp, 'log_evidence': lev}
    if covariance:
        out['spectrum_covariance_diag'] = cov
    return out

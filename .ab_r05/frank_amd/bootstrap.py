"""Bootstrap of a frank fit: the loop of frank/fit.py:731-797 (`perform_bootstrap`) with the table resident on the GPU.

The reference draws N row indices with replacement (`utilities.draw_bootstrap_sample`, utilities.py:632-666), copies
the four columns and runs the whole fit on the copy, `bootstrap_ntrials` times.  Here the table is uploaded once; a
trial's resample is expressed as per-row multiplicities (bincount of the same `np.random.randint` draw, so a seeded
run picks the reference's rows) that the binning pre-pass applies while streaming the table in place -- no gather, no
copy --, and the power-spectrum iteration of trial t runs on its own stream while trial t+1 is being binned.
"""
import ctypes

import numpy as np

from frank_amd import _lib
from frank_amd.radial_fitters import FrankFitter
from frank_amd.statistical_models import GaussianModel, _BAD_P_MSG


def draw_bootstrap_counts(n):
    """Row multiplicities of one bootstrap resample; consumes the global NumPy RNG exactly as
    `draw_bootstrap_sample` does (utilities.py:658)."""
    idxs = np.random.randint(low=0, high=n, size=n)
    return np.bincount(idxs, minlength=n).astype(np.int32)


def bootstrap_fits(fitter, u, v, vis, weights, ntrials, nonnegative=False):
    """`ntrials` bootstrap fits of (u, v, vis, weights) with `fitter` (a frank_amd FrankFitter: hyper-parameters,
    geometry, method).  Returns (r, profiles) with profiles[t] = sol.I of trial t (sol.solve_non_negative() if
    `nonnegative`), what perform_bootstrap saves to `<prefix>_bootstrap.npz` (fit.py:775-783).
    """
    if not isinstance(fitter, FrankFitter):
        raise TypeError("fitter must be a frank_amd FrankFitter")
    L = _lib.lib
    u, v = _lib.f8(u), _lib.f8(v)
    vis = np.asarray(vis)
    Vre = _lib.f8(vis.real)
    Vim = _lib.f8(vis.imag) if np.iscomplexobj(vis) else None
    w = _lib.f8(np.atleast_1d(weights))
    n, N = u.size, fitter.size
    if v.size != n or Vre.size != n or w.size not in (1, n):
        raise ValueError("u, v, V (and weights) must have matching lengths")
    ctx = fitter._DHT.context()
    geom = _lib.make_geometry(fitter._geometry)
    vis_model = _lib.VIS_MODELS[fitter._vis_map._vis_model]
    alpha, p_0, wsmooth, tol = fitter._hyper
    lognormal = fitter._method == 'LogNormal'
    table = ctypes.c_void_p()
    _lib.check(L.fh_vis_upload(fitter._DHT.device, _lib.ptr(u), _lib.ptr(v), _lib.ptr(Vre), _lib.ptr(Vim), _lib.ptr(w), w.size, n,
                               ctypes.byref(table)))
    profiles = np.empty((ntrials, N))
    slots = L.fh_fit_slots()
    pending = []  # (trial, ticket, M, j) of Normal fits in flight

    def collect(entry):
        t, ticket, Mj = entry
        mu, p = np.empty(N), np.empty(N)
        niter = ctypes.c_int(0)
        rc = L.fh_fit_collect(ctx, ticket, _lib.ptr(mu), _lib.ptr(p), ctypes.byref(niter))
        if rc == _lib.FH_ERR_BAD_P:
            raise ValueError(_BAD_P_MSG)
        if rc == _lib.FH_ERR_NOT_SPD:
            # a Cholesky of this trial's loop failed: carry on as the reference does, through the SVD pseudo-inverse
            # (statistical_models.py:747-755), one posterior at a time on this trial's M, j
            keep = fitter._M, fitter._j
            fitter._M, fitter._j = Mj
            try:
                sol = fitter._fit_one_posterior_at_a_time()
            finally:
                fitter._M, fitter._j = keep
            profiles[t] = sol._fit.solve_non_negative() if nonnegative else sol.I
            return
        _lib.check(rc)
        fitter._check_convergence_policy(niter.value)
        if nonnegative:
            fit = GaussianModel._from_solution(fitter._DHT, Mj[0], Mj[1], p, mu)
            profiles[t] = fit.solve_non_negative()
        else:
            profiles[t] = mu

    # the arithmetic of the binning pass is per-context state: set it from THIS fitter, whatever an earlier caller left
    _lib.check(L.fh_ctx_set_arithmetic(ctx, 1 if fitter._vis_map._arithmetic == 'fp32' else 0))
    _lib.check(L.fh_ctx_set_scale_height(
        ctx, _lib.ptr(_lib.f8(fitter._vis_map._H2)) if fitter._vis_map._vis_model == 'debris' else None))
    try:
        for t in range(ntrials):
            counts = draw_bootstrap_counts(n)
            _lib.check(L.fh_vis_set_multiplicity(table, counts.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
            _lib.check(L.fh_bin_reset(ctx))
            _lib.check(L.fh_bin_visibilities(ctx, ctypes.byref(geom), table, 0, n))
            H0, qmn, qmx = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            # M, j of every trial come back (0.7 MB): solve_non_negative needs them, and so does the SVD route the fit
            # continues through when a Cholesky of its loop fails (statistical_models.py:747-755)
            Mj = (np.empty((N, N)), np.empty(N))
            _lib.check(L.fh_stats_finalize(ctx, ctypes.byref(geom), vis_model, 0, _lib.ptr(Mj[0]), _lib.ptr(Mj[1]),
                                           ctypes.byref(H0), ctypes.byref(qmn), ctypes.byref(qmx)))
            if fitter._vis_map.check_qbounds:
                fitter._vis_map._check_uv_range(qmn.value, qmx.value)
            if lognormal:
                s, p = np.empty(N), np.empty(N)
                niter = ctypes.c_int(0)
                _lib.set_lognormal_linesearch(ctx, fitter._lognormal_linesearch)
                rc = L.fh_fit_lognormal(ctx, None, None, alpha, p_0, wsmooth, tol, int(fitter._max_iter),
                                        float(np.exp(fitter._s_scale)), _lib.ptr(s), _lib.ptr(p), ctypes.byref(niter),
                                        None, None, None, None)
                if rc == _lib.FH_ERR_BAD_P:
                    raise ValueError(_BAD_P_MSG)
                _lib.check(rc)
                fitter._check_convergence_policy(niter.value)
                profiles[t] = np.exp(s + fitter._s_scale)
                continue
            if len(pending) == slots:
                collect(pending.pop(0))
            ticket = ctypes.c_int(-1)
            _lib.check(L.fh_fit_submit(ctx, alpha, p_0, wsmooth, tol, int(fitter._max_iter), ctypes.byref(ticket)))
            pending.append((t, ticket.value, Mj))
        _lib.check(L.fh_fit_flush(ctx))  # launch the last, partly filled batch before waiting for the earlier ones
        while pending:
            collect(pending.pop(0))
    finally:
        # an exception above (convergence policy, q range, bad spectrum ...) must not leave tickets outstanding: their
        # fit slots would stay busy on the fitter's long-lived context and starve the next pipeline
        if pending:
            L.fh_fit_flush(ctx)
            scratch_mu, scratch_p, scratch_n = np.empty(N), np.empty(N), ctypes.c_int(0)
            for _t, ticket, _Mj in pending:
                L.fh_fit_collect(ctx, ticket, _lib.ptr(scratch_mu), _lib.ptr(scratch_p), ctypes.byref(scratch_n))
        L.fh_vis_destroy(table)
    return fitter.r, profiles

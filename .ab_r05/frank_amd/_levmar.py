"""Levenberg-Marquardt on the normal equations, for least-squares problems whose residual vector lives on the GPU.

The algorithm is the one MINPACK's lmdif / lmder run -- scipy.optimize.least_squares(method='lm'), which the reference's
geometry fits call (geometry.py:589, :741) -- as More describes it (The Levenberg-Marquardt algorithm: implementation and
theory, 1978): a trust region of radius delta on the scaled step, the parameter `par` found by the More-Hebden iteration so
that |D p| = delta within 10 %, the radius updated from the ratio of actual to predicted reduction, the same three
convergence tests (ftol, xtol, gtol) with SciPy's defaults, D = 1 (least_squares' x_scale=1.0 hands MINPACK diag = 1,
mode = 2).  MINPACK works on a QR factorisation of the m x n Jacobian; here m = 2e7 and n <= 6, the Jacobian never exists on
the host, and everything is phrased through A = J^T J and g = J^T r (R^T R = A, Q^T r restricted to range(J) = R^-T g).
"""
import numpy as np

_EPS = np.finfo(np.float64).eps
_DWARF = np.finfo(np.float64).tiny


def _solve_shifted(A, shift, D, rhs):
    """(A + shift D^2)^-1 rhs, the system scaled to unit diagonal first (the parameters of a geometry differ by 1e4 in size)."""
    B = A + shift * np.diag(D * D)
    s = 1.0 / np.sqrt(np.where(np.diag(B) > 0, np.diag(B), 1.0))
    return s * np.linalg.solve(B * np.outer(s, s), s * rhs)


def _lmpar(A, g, D, delta, par):
    """Levenberg-Marquardt parameter and step: p = -(A + par D^2)^-1 g with |D p| <= 1.1 delta, par = 0 if the Gauss-Newton
    step fits (MINPACK lmpar)."""
    singular = np.any(np.diag(A) <= 0)
    if not singular:
        try:
            np.linalg.cholesky(A * np.outer(1 / np.sqrt(np.diag(A)), 1 / np.sqrt(np.diag(A))))
        except np.linalg.LinAlgError:
            singular = True
    if not singular:
        p = -_solve_shifted(A, 0.0, D, g)
        dxnorm = np.linalg.norm(D * p)
        fp = dxnorm - delta
        if fp <= 0.1 * delta:
            return 0.0, p
        w = D * (D * p) / dxnorm
        parl = fp / (delta * float(w @ _solve_shifted(A, 0.0, D, w)))  # Newton step of phi at par = 0
    else:
        p = np.zeros_like(g)
        dxnorm = 0.0
        fp = -delta
        parl = 0.0
    gnorm = np.linalg.norm(g / D)
    paru = gnorm / delta
    if paru == 0:
        paru = _DWARF / min(delta, 0.1)
    par = min(max(par, parl), paru)
    if par == 0:
        par = gnorm / dxnorm if dxnorm > 0 else paru
    for it in range(1, 11):
        if par == 0:
            par = max(_DWARF, 0.001 * paru)
        p = -_solve_shifted(A, par, D, g)
        dxnorm = np.linalg.norm(D * p)
        prev = fp
        fp = dxnorm - delta
        if abs(fp) <= 0.1 * delta or (parl == 0 and fp <= prev and prev < 0) or it == 10:
            break
        w = D * (D * p) / dxnorm
        parc = fp / (delta * float(w @ _solve_shifted(A, par, D, w)))
        if fp > 0:
            parl = max(parl, par)
        if fp < 0:
            paru = min(paru, par)
        par = max(parl, par + parc)
    return par, p


def levenberg_marquardt(trial, accept, normal_equations, x0, ftol=1e-8, xtol=1e-8, gtol=1e-8, factor=100.0, maxfev=None):
    """Minimise |r(x)|^2.

    trial(x) -> |r(x)|^2 : evaluates the residual at x (it stays wherever the caller keeps trial vectors)
    accept()             : the last trial point becomes the current one
    normal_equations(x)  -> (J^T J, J^T r, evaluations spent) at the current point x
    Returns (x, info, nfev): info as MINPACK's (1-4 converged, 5 too many evaluations, 6-8 tolerances too small).
    """
    x = np.array(x0, dtype=np.float64)
    n = x.size
    maxfev = 100 * n * (n + 1) if maxfev is None else maxfev
    D = np.ones(n)
    fnorm = np.sqrt(trial(x))
    accept()
    nfev, par, it, info = 1, 0.0, 1, 0
    xnorm = delta = 0.0
    while True:
        A, g, spent = normal_equations(x)
        nfev += spent
        acnorm = np.sqrt(np.maximum(np.diag(A), 0.0))
        if it == 1:
            xnorm = np.linalg.norm(D * x)
            delta = factor * xnorm if xnorm != 0 else factor
        gnorm = 0.0
        if fnorm != 0:
            ok = acnorm != 0
            if np.any(ok):
                gnorm = float(np.max(np.abs(g[ok]) / (fnorm * acnorm[ok])))
        if gnorm <= gtol:
            return x, 4, nfev
        while True:
            par, p = _lmpar(A, g, D, delta, par)
            xt = x + p
            pnorm = np.linalg.norm(D * p)
            if it == 1:
                delta = min(delta, pnorm)
            fnorm1 = np.sqrt(trial(xt))
            nfev += 1
            actred = 1.0 - (fnorm1 / fnorm) ** 2 if 0.1 * fnorm1 < fnorm else -1.0
            t1 = np.sqrt(max(float(p @ A @ p), 0.0)) / fnorm if fnorm != 0 else 0.0
            t2 = np.sqrt(par) * pnorm / fnorm if fnorm != 0 else 0.0
            prered = t1 * t1 + t2 * t2 / 0.5
            dirder = -(t1 * t1 + t2 * t2)
            ratio = actred / prered if prered != 0 else 0.0
            if ratio <= 0.25:
                temp = 0.5 if actred >= 0 else 0.5 * dirder / (dirder + 0.5 * actred)
                if 0.1 * fnorm1 >= fnorm or temp < 0.1:
                    temp = 0.1
                delta = temp * min(delta, pnorm / 0.1)
                par /= temp
            elif par == 0 or ratio >= 0.75:
                delta = pnorm / 0.5
                par *= 0.5
            if ratio >= 1e-4:
                x = xt
                accept()
                xnorm = np.linalg.norm(D * x)
                fnorm = fnorm1
                it += 1
            if abs(actred) <= ftol and prered <= ftol and 0.5 * ratio <= 1:
                info = 1
            if delta <= xtol * xnorm:
                info = 3 if info == 1 else 2
            if info:
                return x, info, nfev
            if nfev >= maxfev:
                info = 5
            if abs(actred) <= _EPS and prered <= _EPS and 0.5 * ratio <= 1:
                info = 6
            if delta <= _EPS * xnorm:
                info = 7
            if gnorm <= _EPS:
                info = 8
            if info:
                return x, info, nfev
            if ratio >= 1e-4:
                break


def forward_steps(x):
    """Step of MINPACK's forward-difference Jacobian (fdjac2 with epsfcn = 0): sqrt(eps) |x_j|, sqrt(eps) where x_j = 0."""
    h = np.sqrt(_EPS) * np.abs(x)
    return np.where(h == 0, np.sqrt(_EPS), h)

"""CriticalFilter -- drop-in for frank/filter.py:23-263.

`update_power_spectrum` runs one device pass (fh_update_power_spectrum: rocBLAS/rocSOLVER solve + the
fit_update kernel).  FrankFitter does not call it per iteration: the whole loop runs inside
fh_fit_normal.  `spectral_smoothing_matrix` is the host mirror of filter.py:23-62 (dense, O(N)).
"""
import numpy as np

from frank_amd import _lib


def spectral_smoothing_matrix(DHT, weights):
    r"""T = weights * Delta^T diag(dce) Delta in log q (filter.py:23-62), returned dense (N x N)."""
    log_q = np.log(DHT.q)
    dc = (log_q[2:] - log_q[:-2]) / 2
    de = np.diff(log_q)
    N = DHT.size
    Delta = np.zeros([N, N])
    i = np.arange(1, N - 1)
    Delta[i, i - 1] = 1 / (dc * de[:-1])
    Delta[i, i] = -(1 / de[1:] + 1 / de[:-1]) / dc
    Delta[i, i + 1] = 1 / (dc * de[1:])
    dce = np.zeros_like(log_q)
    dce[1:-1] = dc
    return weights * Delta.T.dot(dce[:, None] * Delta)


class CriticalFilter:
    """Optimizer for power-spectrum priors (filter.py:64-263)."""

    def __init__(self, DHT, alpha, p_0, weights_smooth, tol=1e-3):
        self._DHT = DHT
        self._alpha = alpha
        self._p_0 = p_0
        self._rho = 1.0
        self._tol = tol
        self._weights_smooth = weights_smooth
        self._Tij = spectral_smoothing_matrix(DHT, weights_smooth)

    def update_power_spectrum(self, fit):
        """Estimate the best fit power spectrum given the current model fit (filter.py:154-177)."""
        N = self._DHT.size
        p_new = np.empty(N)
        if hasattr(fit, '_Dinv'):
            # a posterior that carries its own precision matrix (LogNormalMAPModel): Tr1 from fit.MAP, Tr2 from Dinv^-1
            _lib.check(_lib.lib.fh_posterior_update(
                self._DHT.context(), _lib.ptr(_lib.f8(fit.MAP)), _lib.ptr(_lib.f8(fit._Dinv)),
                _lib.ptr(_lib.f8(fit.power_spectrum)), float(self._alpha), float(self._p_0),
                float(self._weights_smooth), _lib.ptr(p_new)))
            return p_new
        rc = _lib.lib.fh_update_power_spectrum(
            self._DHT.context(), _lib.ptr(_lib.f8(fit._M)), _lib.ptr(_lib.f8(fit._j)),
            _lib.ptr(_lib.f8(fit.power_spectrum)), float(self._alpha), float(self._p_0),
            float(self._weights_smooth), None, _lib.ptr(p_new))
        if rc == _lib.FH_ERR_NOT_SPD:
            return self._update_through_dsolve(fit)
        _lib.check(rc)
        return p_new

    def _update_through_dsolve(self, fit):
        """filter.py:154-177 spelled out with the posterior's own solves -- taken when the Cholesky of M + S^-1 fails and
        `fit` went through the SVD pseudo-inverse (statistical_models.py:747-755): Tr1 = (Y mu)^2, Tr2_i = sum_j Y_ij
        [D Y^T]_ji with D.b from fit.Dsolve (device SVD route), then the pentadiagonal solve on the host (O(N^3) dense
        here: this path is the exception, not the loop)."""
        Ykm = self._DHT.coefficients()
        pi = fit.power_spectrum
        Tr1 = np.dot(Ykm, fit.MAP) ** 2
        Tr2 = np.einsum('ij,ji->i', Ykm, fit.Dsolve(Ykm.T))
        beta = (self._p_0 + 0.5 * (Tr1 + Tr2)) / pi - (self._alpha - 1.0 + 0.5 * 1.0)
        Tij_pI = np.asarray(self._Tij) + np.eye(self._DHT.size)
        tau = np.linalg.solve(Tij_pI, beta + np.log(pi))
        return np.exp(tau)

    def check_convergence(self, pi_new, pi_old):
        """filter.py:179-181"""
        return np.all(np.abs(pi_new - pi_old) <= self._tol * pi_new)

    def covariance_MAP(self, fit, ret_inv=False):
        """Covariance of the power spectrum at maximum likelihood (filter.py:184-227)."""
        Ykm = self._DHT.coefficients()
        mq = np.dot(Ykm, fit.MAP)
        mqq = np.outer(mq, mq)
        Dqq = np.dot(Ykm, np.dot(fit.covariance, Ykm.T))
        p = fit.power_spectrum
        hess = (np.diag(self._p_0 / p + 0.5 * (mq ** 2 + np.diag(Dqq)) / p) + self._Tij
                - 0.5 * np.outer(1 / p, 1 / p) * (2 * mqq + Dqq) * Dqq)
        if ret_inv:
            return hess
        return np.linalg.inv(hess)

    def log_prior(self, p):
        r"""log P(p) up to a constant (filter.py:229-263)."""
        xi = self._p_0 / p
        like = - np.sum(xi + (self._alpha - 1) * np.log(xi))
        tau = np.log(p)
        like -= 0.5 * np.dot(tau, self._Tij.dot(tau))
        return like

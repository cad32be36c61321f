"""DiscreteHankelTransform -- drop-in for frank/hankel.py:25-294 (nu = 0) on top of libfrank_hip.

Set-up (collocation points, Ykm, scale factors: hankel.py:55-93) is done by the C ABI on the host
(fh_dht_create); every evaluation at user-supplied points q (hankel.py:201-202) runs on the GPU
(fh_dht_coefficients).  Instances hold a native handle but pickle as plain numbers (it is rebuilt).
"""
import ctypes
import os

import numpy as np

from frank_amd import _lib


def default_device():
    """The HIP device used when none is named: $FRANK_AMD_DEVICE, else 0."""
    return int(os.environ.get("FRANK_AMD_DEVICE", "0"))


class DiscreteHankelTransform(object):
    r"""Utilities for computing the discrete Hankel transform (same contract as the reference class).

    Parameters
    ----------
    Rmax : float
        Maximum radius beyond which f(r) is zero (radians at this level, radial_fitters.py:441)
    N : integer
        Number of terms to use in the series
    nu : integer, default = 0
        Order of the Bessel function; only nu = 0 is built.
    device : integer, optional (not in the reference)
        HIP device that carries this transform's GPU work.  Default: $FRANK_AMD_DEVICE, else 0.  One process
        may hold transforms on several devices (frank_amd.sweep.sweep_fits(..., devices=[...])).
    """

    def __init__(self, Rmax, N, nu=0, device=None):
        if nu != 0:
            raise NotImplementedError("frank_amd builds the nu = 0 transform only (hankel.py:58-66)")
        self._N = int(N)
        self._nu = nu
        self._Rmax = float(Rmax)
        self._device = default_device() if device is None else int(device)
        self._handle = None
        self._ctx = None
        self._build()

    # -- native handle management ---------------------------------------------------------------------
    def _build(self):
        h = ctypes.c_void_p()
        _lib.check(_lib.lib.fh_dht_create(self._Rmax, self._N, self._nu, ctypes.byref(h)))
        self._handle = h
        N = self._N
        self._Rnk, self._Qnk = np.empty(N), np.empty(N)
        zeros = np.empty(N + 1)
        self._Ykm = np.empty((N, N))
        self._scale_factor = np.empty(N)
        Qmax, Rmax = ctypes.c_double(), ctypes.c_double()
        _lib.check(_lib.lib.fh_dht_get(h, _lib.ptr(self._Rnk), _lib.ptr(self._Qnk), _lib.ptr(zeros),
                                       _lib.ptr(self._Ykm), _lib.ptr(self._scale_factor), ctypes.byref(Qmax),
                                       ctypes.byref(Rmax)))
        self._j_nk, self._j_nN = zeros[:-1].copy(), float(zeros[-1])
        self._Qmax = Qmax.value

    @property
    def device(self):
        """HIP device of this transform's GPU work"""
        return self._device

    def context(self, device=None):
        """The fh_ctx (device buffers + stream) of this transform on `device` (default: its own device); created
        on first use.  A transform keeps one context per device it has been used on."""
        device = self._device if device is None else int(device)
        if self._ctx is None:
            self._ctx = {}
        c = self._ctx.get(device)
        if c is None:
            c = ctypes.c_void_p()
            _lib.check(_lib.lib.fh_ctx_create(self._handle, device, ctypes.byref(c)))
            _lib.warn_if_any()
            self._ctx[device] = c
        return c

    def _drop_ctx(self):
        ctxs = getattr(self, "_ctx", None)
        if ctxs:
            for c in ctxs.values():
                _lib.lib.fh_ctx_destroy(c)
        self._ctx = None

    def __del__(self):
        try:
            self._drop_ctx()
            if getattr(self, "_handle", None) is not None:
                _lib.lib.fh_dht_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def __getstate__(self):
        return dict(Rmax=self._Rmax, N=self._N, nu=self._nu, device=self._device)

    def __setstate__(self, state):
        self._Rmax, self._N, self._nu = state["Rmax"], state["N"], state["nu"]
        self._device = state.get("device", default_device())
        self._handle = None
        self._ctx = None
        self._build()

    # -- reference API ----------------------------------------------------------------------------------
    @classmethod
    def get_collocation_points(cls, Rmax, N, nu=0):
        """hankel.py:95-125"""
        d = cls(Rmax, N, nu)
        return d.r, d.q

    def transform(self, f, q=None, direction='forward'):
        """hankel.py:127-165"""
        if q is None:
            Y = self._Ykm
            if direction == 'forward':
                norm = (2 * np.pi * self._Rmax ** 2) / self._j_nN
            elif direction == 'backward':
                norm = (2 * np.pi * self._Qmax ** 2) / self._j_nN
            else:
                raise AttributeError("direction must be one of {}".format(['forward', 'backward']))
        else:
            Y = self.coefficients(q, direction=direction)
            norm = 1.0
        return norm * np.dot(Y, f)

    def coefficients(self, q=None, direction='forward'):
        """hankel.py:167-204.  With q given, H = (norm*scale_factor) * J0(outer(k q, j_nk)) is built on the GPU."""
        if direction == 'forward':
            norm = 1 / (np.pi * self._Qmax ** 2)
        elif direction == 'backward':
            norm = 1 / (np.pi * self._Rmax ** 2)
        else:
            raise AttributeError("direction must be one of {}".format(['forward', 'backward']))
        if q is None:
            return 0.5 * self._j_nN * norm * self._Ykm
        return self._device_coefficients(q, direction, 1.0)

    def interpolation_coefficients(self, q, space='Real'):
        """hankel.py:206-236: the matrix Y with f(q) = np.dot(Y, f) of the Fourier-Bessel interpolation,
        Y[i, k] = J0(x_i) [x_i < j_N] * 2 j_k / J1(j_k) / (j_k^2 - x_i^2), x = 2 pi q Qmax ('Real') or 2 pi q Rmax ('Fourier').
        J0(x_i) comes from the device generator of `coefficients` (its first column evaluated at q_i = x_i Qmax / j_1), J1(j_k)
        from the scale factor 1 / J1(j_k)^2 the transform already holds (the sign alternates from zero to zero)."""
        if space == 'Real':
            x = np.atleast_1d(2 * np.pi * np.asarray(q, dtype=float) * self._Qmax)
        elif space == 'Fourier':
            x = np.atleast_1d(2 * np.pi * np.asarray(q, dtype=float) * self._Rmax)
        else:
            raise ValueError("Space must be one of 'Real' or 'Fourier', not "
                             f"{space}.")
        x = x.reshape(-1)
        norm = 1 / (np.pi * self._Qmax ** 2)
        H = self._device_coefficients(x * (self._Qmax / self._j_nk[0]), 'forward', 1.0)
        j0x = H[:, 0] / (norm * self._scale_factor[0])
        jnup = np.where(np.arange(self._N) % 2 == 0, 1.0, -1.0) / np.sqrt(self._scale_factor)  # J1 at the zeros of J0
        coeff = np.outer(np.where(x < self._j_nN, j0x, 0), 2 * self._j_nk / jnup)
        return coeff / (self._j_nk.reshape(1, -1) ** 2 - x.reshape(-1, 1) ** 2)

    def interpolate(self, f, q, space='Real'):
        """hankel.py:238-263: f (given at the collocation points) at the points q, consistent with the Fourier-Bessel series."""
        return np.dot(self.interpolation_coefficients(q, space), f)

    def _device_coefficients(self, q, direction, scale):
        q = _lib.f8(np.atleast_1d(q)).reshape(-1)
        H = np.empty((q.size, self._N))
        _lib.check(_lib.lib.fh_dht_coefficients(self.context(), _lib.ptr(q), q.size,
                                                0 if direction == 'forward' else 1, float(scale), _lib.ptr(H)))
        return H

    @property
    def r(self):
        """Radius points"""
        return self._Rnk

    @property
    def Rmax(self):
        """Maximum radius"""
        return self._Rmax

    @property
    def q(self):
        """Frequency points"""
        return self._Qnk

    @property
    def Qmax(self):
        """Maximum frequency"""
        return self._Qmax

    @property
    def size(self):
        """Number of points used in the DHT"""
        return self._N

    @property
    def order(self):
        """Order of the Bessel function"""
        return self._nu

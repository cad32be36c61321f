"""Fixed source geometry -- host mirror of frank/geometry.py:41-131, 173-401.

The hot path does NOT use these NumPy routines: VisibilityMapping.map_visibilities hands the raw
(u, v, V) to the bin_gram kernel, which applies the phase shift and deprojection itself
(csrc/bin_gram.hip::deproject_one).  They exist for callers that need deprojected coordinates on the
host (FrankRadialFit.predict, user code) and carry the (inc, PA, dRA, dDec) the kernel reads.
Geometry *fitting* (FitGeometryGaussian / FitGeometryFourierBessel, geometry.py:404-763) is out of scope.
"""
import numpy as np

from frank_amd.constants import rad_to_arcsec, deg_to_rad


def apply_phase_shift(u, v, V, dRA, dDec, inverse=False):
    """geometry.py:41-79"""
    dRA = dRA * (2. * np.pi / rad_to_arcsec)
    dDec = dDec * (2. * np.pi / rad_to_arcsec)
    phi = u * dRA + v * dDec
    if inverse:
        return V / (np.cos(phi) + 1j * np.sin(phi))
    return V * (np.cos(phi) + 1j * np.sin(phi))


def deproject(u, v, inc, PA, inverse=False):
    """geometry.py:82-131"""
    inc = inc * deg_to_rad
    PA = PA * deg_to_rad
    cos_t = np.cos(PA)
    sin_t = np.sin(PA)
    if inverse:
        sin_t *= -1
        u = u / np.cos(inc)
    up = u * cos_t - v * sin_t
    vp = u * sin_t + v * cos_t
    if inverse:
        return up, vp
    wp = up * np.sin(inc)
    up = up * np.cos(inc)
    return up, vp, wp


class SourceGeometry(object):
    """geometry.py:173-369 (inc, PA in degrees; dRA, dDec in arcsec)."""

    def __init__(self, inc=None, PA=None, dRA=None, dDec=None):
        self._inc = inc
        self._PA = PA
        self._dRA = dRA
        self._dDec = dDec

    def apply_correction(self, u, v, V, use3D=False):
        Vp = apply_phase_shift(u, v, V, self._dRA, self._dDec, inverse=True)
        up, vp, wp = deproject(u, v, self._inc, self._PA)
        if use3D:
            return up, vp, wp, Vp
        return up, vp, Vp

    def undo_correction(self, u, v, V):
        up, vp = self.reproject(u, v)
        Vp = apply_phase_shift(up, vp, V, self._dRA, self._dDec, inverse=False)
        return up, vp, Vp

    def deproject(self, u, v, use3D=False):
        if use3D:
            return deproject(u, v, self._inc, self._PA)
        return deproject(u, v, self._inc, self._PA)[:2]

    def reproject(self, u, v):
        return deproject(u, v, self._inc, self._PA, inverse=True)

    def fit(self, u, v, V, weights):
        return

    def clone(self):
        return FixedGeometry(self.inc, self.PA, self.dRA, self.dDec)

    @property
    def dRA(self):
        return self._dRA

    @property
    def dDec(self):
        return self._dDec

    @property
    def PA(self):
        return self._PA

    @property
    def inc(self):
        return self._inc

    @property
    def rescale_factor(self):
        return 1.0 / np.cos(self._inc * deg_to_rad)

    def __repr__(self):
        return "SourceGeometry(inc={}, PA={}, dRA={}, dDec={})".format(self.inc, self.PA, self.dRA, self.dDec)


class FixedGeometry(SourceGeometry):
    """geometry.py:372-401"""

    def __init__(self, inc, PA, dRA=0.0, dDec=0.0):
        super(FixedGeometry, self).__init__(inc, PA, dRA, dDec)

    def __repr__(self):
        return "FixedGeometry(inc={}, PA={}, dRA={}, dDEC={})".format(self.inc, self.PA, self.dRA, self.dDec)

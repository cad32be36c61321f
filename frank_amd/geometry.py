"""Fixed source geometry (inclination, position angle, phase centre) for the MI355X frank path.

API-compatible with the fixed-geometry part of frank/geometry.py (`apply_phase_shift` :41-79, `deproject`
:82-131, `SourceGeometry` :173-369, `FixedGeometry` :372-401).  The hot path does NOT run these NumPy
routines: `VisibilityMapping.map_visibilities` hands the raw (u, v, V) to the GPU, where
`deproject_kernel` (csrc/bin_gram.hip) applies the phase shift and the deprojection.  They serve callers that
need deprojected coordinates on the host (`FrankRadialFit.predict`, user code) and carry the
(inc, PA, dRA, dDec) that the kernels read.  Geometry *fitting* (geometry.py:404-763) is out of scope.
"""
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

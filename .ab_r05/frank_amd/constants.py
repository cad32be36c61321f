"""Unit conversions used by the drop-in classes.

The three names and their values are those of the reference (frank/constants.py:23-25); the values must be the
same IEEE doubles because Rmax [arcsec] / rad_to_arcsec defines the collocation points.  They are spelled through
`math` here and checked bit-for-bit against the expressions NumPy would evaluate in tests/test_host_api.py.
"""
import math

_ARCSEC_PER_DEGREE = 3600
_DEGREES_PER_HALF_TURN = 180

#: radians -> arcseconds (648000 / pi)
rad_to_arcsec = _ARCSEC_PER_DEGREE * _DEGREES_PER_HALF_TURN / math.pi
#: steradians -> square arcseconds
sterad_to_arcsec = rad_to_arcsec * rad_to_arcsec
#: degrees -> radians
deg_to_rad = math.pi / _DEGREES_PER_HALF_TURN

__all__ = ["rad_to_arcsec", "sterad_to_arcsec", "deg_to_rad"]

"""Conversion constants, bit-identical to frank/constants.py:23-25."""
import numpy as np

rad_to_arcsec = 3600 * 180 / np.pi
sterad_to_arcsec = rad_to_arcsec ** 2
deg_to_rad = np.pi / 180

#!/usr/bin/env python3
"""Generate frank_amd/mock_disc_vis_table.npz (run in the BUILD container only).

Imports the reference (/root/reference, frank v1.2.3) and tabulates the noiseless,
deprojected visibility curve of the mock disc of docs/tutorials/mock_data.ipynb
(cells 8, 16-17) exactly as frank.utilities.make_mock_data computes it
(utilities.py:962-1038 -> generic_dht, N=500, Rmax=2 arcsec, x cos(inc)).
frank_amd/mock.py interpolates this table, so 1e7-1e8 synthetic visibilities can be
drawn in seconds on any box without the reference.
"""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
from frank.utilities import generic_dht  # noqa: E402

INC = 34.97
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "frank_amd", "mock_disc_vis_table.npz")


def gauss(x, a, x0, sigma):
    return a * np.exp(-(x - x0) ** 2 / (2 * sigma ** 2))


def sigma(fwhm):
    return fwhm / (8 * np.sqrt(np.log(2)))


def main():
    r = np.linspace(0, 0.7, 1000)
    I = (gauss(r, 1e10, 0.0, sigma(1.0)) - gauss(r, 5e9, 0.1, sigma(0.2))
         - gauss(r, 1e9, 0.3, sigma(0.2)) + gauss(r, 2e9, 0.5, sigma(0.1)))
    q = np.linspace(0.0, 2.4e6, 16385)
    _, V = generic_dht(r, I, Rmax=2.0, N=500, grid=q, inc=INC)
    np.savez_compressed(OUT, q=q, V=V, inc=INC, r=r, I=I)
    print("wrote", OUT, V[:3], V.shape)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""UVDataBinner at 1e7 rows: run under `rocprofv3 --kernel-trace --stats` for the kernel times (development tool)."""
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd.utilities import UVDataBinner

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 7
rng = np.random.default_rng(0)
q = np.exp(rng.uniform(np.log(1e4), np.log(2e6), n))
V = rng.normal(size=n) + 1j * rng.normal(size=n)
w = rng.uniform(0.5, 2.0, n)
for bw in (2e4, 1e3):
    for rep in range(3):
        t = time.perf_counter()
        b = UVDataBinner(q, V, w, bw)
        dt = time.perf_counter() - t
    from frank_amd import _lib
    kms = _lib.lib.fh_uvbin_kernel_ms(b._handle)
    # max pass reads 8 B/row, the sum and error passes 32 B/row each
    print("n=%d bin_width=%g: %d bins, %.1f ms per UVDataBinner incl. upload; kernels %.3f ms = %.0f GB/s of 72 B/row" % (
        n, bw, len(b), dt * 1e3, kms, 72.0 * n / kms / 1e6))

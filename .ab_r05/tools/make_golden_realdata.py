#!/usr/bin/env python3
"""Real-data fixture (run in the BUILD container only; imports the reference).

    python3 tools/make_golden_realdata.py

Input: docs/tutorials/multi_ring_C43_6.txt.bz2 of the reference tree -- a simulated ALMA C43-6 observation of a
multi-ring disc (54 180 visibilities, unit weights), the only complete uv-table the reference ships (the AS 209
blob its tests use is absent).  The table itself is DATA and is stored in the fixture next to what the reference
computes from it: M, j, H0 (statistical_models.py:109-237) and the FrankFitter result (radial_fitters.py:737-832)
for N=100, Rmax=2", face-on geometry, alpha=1.05, w_smooth=1e-4.
"""
import bz2
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, ".."))
sys.path.insert(0, "/root/reference")

import scipy  # noqa: E402
import frank  # noqa: E402
from frank.geometry import FixedGeometry  # noqa: E402
from frank.radial_fitters import FrankFitter  # noqa: E402

SRC = "/root/reference/docs/tutorials/multi_ring_C43_6.txt.bz2"


def main():
    with bz2.open(SRC, "rt") as f:
        tab = np.loadtxt(f)
    u, v, re, im, w = tab.T
    V = re + 1j * im
    geom = dict(inc=0.0, PA=0.0, dRA=0.0, dDec=0.0)
    FF = FrankFitter(2.0, 100, FixedGeometry(**geom), alpha=1.05, weights_smooth=1e-4,
                     store_iteration_diagnostics=True, verbose=False)
    t0 = time.perf_counter()
    m = FF.preprocess_visibilities(u, v, V, w)
    sol = FF.fit_preprocessed(m)
    dt = time.perf_counter() - t0
    d = FF.iteration_diagnostics
    print("N=100 fit of %d real-format visibilities: niter=%d (%.1f s)" % (u.size, d["num_iterations"], dt))
    q_pred = np.array([2e4, 1e5, 4e5, 1.5e6])
    out = dict(u=u, v=v, Vre=re, Vim=im, w=w, N=100, Rmax=2.0, alpha=1.05, wsmooth=1e-4, M=m["M"], j=m["j"],
               H0=m["null_likelihood"], I=sol.I, p=sol.power_spectrum, niter=d["num_iterations"],
               diag_p_first=np.array(d["power_spectrum"][:3]), q_pred=q_pred, Vpred=sol.predict_deprojected(q_pred),
               meta_reference_version=frank.__version__, meta_numpy=np.__version__, meta_scipy=scipy.__version__,
               **{"geom_" + k: val for k, val in geom.items()})
    path = os.path.join(ROOT, "tests", "golden", "realdata_multi_ring_N100.npz")
    np.savez_compressed(path, **out)
    print("wrote %s %.1f KB" % (path, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()

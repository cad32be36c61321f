#!/usr/bin/env python3
"""Fixture for the mock-data helpers (run in the BUILD container only; imports the reference).

    python3 tools/make_golden_mockdata.py

utilities.generic_dht (forward on the collocation points and on a grid, backward), make_mock_data (no projection;
deprojected with noise, seeded), add_vis_noise (real and complex), get_collocation_points of the reference on a small
analytic profile; inputs stored next to the outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, ".."))
sys.path.insert(0, "/root/reference")

import scipy  # noqa: E402
import frank  # noqa: E402
from frank.geometry import FixedGeometry  # noqa: E402
from frank.utilities import add_vis_noise, generic_dht, get_collocation_points, make_mock_data  # noqa: E402


def main():
    r = np.linspace(0, 1.5, 400)
    I = 1e10 * np.exp(-0.5 * (r / 0.3) ** 2) + 3e9 * np.exp(-0.5 * ((r - 0.8) / 0.1) ** 2)
    rng = np.random.default_rng(9)
    q = np.exp(rng.uniform(np.log(1e4), np.log(1.5e6), 500))
    th = rng.uniform(0, 2 * np.pi, 500)
    u, v = q * np.cos(th), q * np.sin(th)
    w = rng.uniform(100.0, 900.0, 500)
    out = dict(r=r, I=I, u=u, v=v, w=w, Rmax=2.0, N=120)
    out["fwd_grid"], out["fwd"] = generic_dht(r, I, 2.0, 120)
    _, out["fwd_on_q"] = generic_dht(r, I, 2.0, 120, grid=q, inc=40.0)
    out["bwd_grid"], out["bwd"] = generic_dht(out["fwd_grid"], out["fwd"], 2.0, 120, direction="backward")
    rr = np.linspace(0.01, 1.9, 57)
    _, out["bwd_on_r"] = generic_dht(out["fwd_grid"], out["fwd"], 2.0, 120, direction="backward", grid=rr, inc=40.0)
    out["rr"] = rr
    out["mock_plain_q"], out["mock_plain"] = make_mock_data(r, I, 2.0, u, v, N=120)
    g = FixedGeometry(40.0, 70.0, 0.0, 0.0)
    out["mock_deproj_q"], out["mock_deproj"] = make_mock_data(r, I, 2.0, u, v, projection="deproject", geometry=g, N=120,
                                                              add_noise=True, weights=w, seed=17)
    out["mock_reproj_q"], out["mock_reproj"] = make_mock_data(r, I, 2.0, u, v, projection="reproject", geometry=g, N=120)
    Vc = out["mock_plain"] * np.exp(1j * 0.3)
    out["noise_complex_in"] = Vc
    out["noise_complex"] = add_vis_noise(Vc, w, seed=5)
    out["noise_real"] = add_vis_noise(out["mock_plain"], w, seed=5)
    out["coll_r"] = get_collocation_points(2.0, 120)
    out["coll_q"] = get_collocation_points(2.0, 120, direction="backward")
    out.update(meta_reference_version=frank.__version__, meta_numpy=np.__version__, meta_scipy=scipy.__version__)
    path = os.path.join(ROOT, "tests", "golden", "mockdata_helpers.npz")
    np.savez_compressed(path, **out)
    print("wrote %s %.1f KB" % (path, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()

"""Pipelined fits (fh_fit_submit / collect) and batched sweeps (fh_fit_normal_batched) against synchronous fits over several basis
sizes: iteration counts equal, profiles to 1e-9.   python3 tools/size_sweep_pipeline.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import FixedGeometry, FrankFitter, _lib  # noqa: E402
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities  # noqa: E402
from frank_amd.sweep import sweep_fits  # noqa: E402

L = _lib.lib
u, v, V, w = mock_disc_visibilities(60000, seed=31, noise_seed=32)
bad = []
for N in (17, 50, 100, 127, 255, 300, 320, 321, 400, 511, 600, 639):
    kw = dict(alpha=1.3, weights_smooth=1e-2, verbose=False, check_qbounds=False, store_iteration_diagnostics=True)
    FF = FrankFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), **kw)
    pre = FF.preprocess_visibilities(u, v, V, w)
    sol = FF.fit_preprocessed(pre)
    nit = FF.iteration_diagnostics["num_iterations"]
    # batched sweep: two points, the first the same hyper-parameters
    sols, its = sweep_fits(FF, pre, np.array([1.3, 1.2]), np.array([1e-2, 1e-1]))
    e_sw = np.abs(sols[0].I - sol.I).max() / np.abs(sol.I).max()
    # pipeline: 5 submissions of the same table
    ctx = FF._DHT.context()
    vis = ctypes.c_void_p()
    Vre, Vim = np.ascontiguousarray(V.real), np.ascontiguousarray(V.imag)
    _lib.check(L.fh_vis_upload(0, _lib.ptr(u), _lib.ptr(v), _lib.ptr(Vre), _lib.ptr(Vim), _lib.ptr(w), w.size, u.size, ctypes.byref(vis)))
    gm = _lib.make_geometry(FixedGeometry(**MOCK_GEOMETRY))
    tickets = []
    for i in range(5):
        _lib.check(L.fh_bin_reset(ctx))
        _lib.check(L.fh_bin_visibilities(ctx, ctypes.byref(gm), vis, 0, u.size))
        _lib.check(L.fh_stats_finalize(ctx, ctypes.byref(gm), 0, 0, None, None, None, None, None))
        t = ctypes.c_int(-1)
        _lib.check(L.fh_fit_submit(ctx, 1.3, 1e-15, 1e-2, 1e-3, 2000, ctypes.byref(t)))
        tickets.append(t.value)
    _lib.check(L.fh_fit_flush(ctx))
    e_pl, n_pl = 0.0, set()
    for t in tickets:
        mu, p, k = np.empty(N), np.empty(N), ctypes.c_int()
        _lib.check(L.fh_fit_collect(ctx, t, _lib.ptr(mu), _lib.ptr(p), ctypes.byref(k)))
        e_pl = max(e_pl, np.abs(mu - sol.I).max() / np.abs(sol.I).max())
        n_pl.add(k.value)
    L.fh_vis_destroy(vis)
    ok = its[0] == nit and e_sw < 1e-9 and n_pl == {nit} and e_pl < 1e-9
    if not ok:
        bad.append(N)
    print("N=%3d  niter %d | sweep niter %d err %.1e | pipeline niter %s err %.1e %s" % (N, nit, its[0], e_sw, sorted(n_pl), e_pl, "" if ok else " <-- MISMATCH"), flush=True)
print("mismatches:", bad)

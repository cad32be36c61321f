"""Soak of the cluster mode: many single fits and shallow pipelines at several basis sizes, every result compared with the first
(same bits expected), cluster fall-backs counted.   timeout 900 python tools/soak_cluster.py [rounds]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import FixedGeometry, FrankFitter, _lib  # noqa: E402
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities  # noqa: E402

L = _lib.lib
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
u, v, V, w = mock_disc_visibilities(200000, seed=7, noise_seed=8)
bad = 0
t00 = time.time()
for N in (130, 200, 300, 320, 335, 400):
    FF = FrankFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), verbose=False, check_qbounds=False)
    pre = FF.preprocess_visibilities(u, v, V, w)
    M, j = np.ascontiguousarray(pre["M"]), np.ascontiguousarray(pre["j"])
    ctx = FF._DHT.context()
    ref = None
    t0 = time.time()
    for r in range(rounds):
        mu, p, nit = np.empty(N), np.empty(N), ctypes.c_int()
        _lib.check(L.fh_fit_normal(ctx, _lib.ptr(M), _lib.ptr(j), 1.3, 1e-15, 1e-2, 1e-3, 300, _lib.ptr(mu), _lib.ptr(p), ctypes.byref(nit), None, None))
        if ref is None:
            ref = (mu.copy(), p.copy(), nit.value)
        elif not (np.array_equal(mu, ref[0]) and np.array_equal(p, ref[1]) and nit.value == ref[2]):
            bad += 1
    # shallow pipelines: 24 fits outstanding, on clusters
    _lib.check(L.fh_stats_upload(ctx, _lib.ptr(M), _lib.ptr(j))) if hasattr(L, "fh_stats_upload") else None
    for r in range(max(1, rounds // 10)):
        tickets = []
        for k in range(24):
            t = ctypes.c_int(-1)
            _lib.check(L.fh_fit_submit(ctx, 1.3, 1e-15, 1e-2, 1e-3, 300, ctypes.byref(t)))
            tickets.append(t.value)
        for t in tickets:
            mu, p, nit = np.empty(N), np.empty(N), ctypes.c_int()
            _lib.check(L.fh_fit_collect(ctx, t, _lib.ptr(mu), _lib.ptr(p), ctypes.byref(nit)))
            if not (np.array_equal(mu, ref[0]) and np.array_equal(p, ref[1]) and nit.value == ref[2]):
                bad += 1
    wg, fb = ctypes.c_int(0), ctypes.c_int64(0)
    _lib.check(L.fh_fit_cluster_info(ctx, ctypes.byref(wg), ctypes.byref(fb)))
    print("N = %d: %d single fits + %d pipelined, %d passes each, %.1f s; workgroups %d, fall-backs %d, differing results so far %d" % (
        N, rounds, 24 * max(1, rounds // 10), ref[2], time.time() - t0, wg.value, fb.value, bad), flush=True)
print("total %.1f s, differing results: %d" % (time.time() - t00, bad))

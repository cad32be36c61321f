"""Soak of the register-resident fit loop: deep pipelines (240 fits in flight, the form taken by load and forced) and batched
launches of 256 identical fits at several basis sizes, every result compared with a fit by the forms that work in memory (same
bits expected).   timeout 900 python3 tools/soak_rr.py [pipelined fits per size]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frank_amd import FixedGeometry, FrankFitter, _lib  # noqa: E402
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities  # noqa: E402

L = _lib.lib
npipe = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
u, v, V, w = mock_disc_visibilities(200000, seed=7, noise_seed=8)
bad = 0
t00 = time.time()
for N in (300, 130, 200, 255, 303):
    FF = FrankFitter(2.0, N, FixedGeometry(**MOCK_GEOMETRY), verbose=False, check_qbounds=False)
    pre = FF.preprocess_visibilities(u, v, V, w)
    M, j = np.ascontiguousarray(pre["M"]), np.ascontiguousarray(pre["j"])
    ctx = FF._DHT.context()
    os.environ["FRANK_AMD_K2_RR"] = "0"
    os.environ["FRANK_AMD_K2_CLUSTER"] = "1"
    mu0, p0, nit0 = np.empty(N), np.empty(N), ctypes.c_int()
    _lib.check(L.fh_fit_normal(ctx, _lib.ptr(M), _lib.ptr(j), 1.3, 1e-15, 1e-2, 1e-3, 300, _lib.ptr(mu0), _lib.ptr(p0), ctypes.byref(nit0), None, None))
    del os.environ["FRANK_AMD_K2_CLUSTER"]
    t0 = time.time()
    for mode in (None, "1"):
        if mode is None:
            os.environ.pop("FRANK_AMD_K2_RR", None)
        else:
            os.environ["FRANK_AMD_K2_RR"] = mode
        _lib.check(L.fh_stats_upload(ctx, _lib.ptr(M), _lib.ptr(j)))
        slots = L.fh_fit_slots()
        pend = []

        def collect(t):
            global bad
            mu, p, nit = np.empty(N), np.empty(N), ctypes.c_int()
            _lib.check(L.fh_fit_collect(ctx, t, _lib.ptr(mu), _lib.ptr(p), ctypes.byref(nit)))
            if not (np.array_equal(mu, mu0) and np.array_equal(p, p0) and nit.value == nit0.value):
                bad += 1
        for k in range(npipe):
            if len(pend) == slots:
                collect(pend.pop(0))
            t = ctypes.c_int(-1)
            _lib.check(L.fh_fit_submit(ctx, 1.3, 1e-15, 1e-2, 1e-3, 300, ctypes.byref(t)))
            pend.append(t.value)
        _lib.check(L.fh_fit_flush(ctx))
        for t in pend:
            collect(t)
        B = 256
        al, pz, ws = np.full(B, 1.3), np.full(B, 1e-15), np.full(B, 1e-2)
        mub, pb = np.empty((B, N)), np.empty((B, N))
        nb_, st = (ctypes.c_int * B)(), (ctypes.c_int * B)()
        for r in range(3):
            _lib.check(L.fh_fit_normal_batched(ctx, _lib.ptr(M), _lib.ptr(j), B, _lib.ptr(al), _lib.ptr(pz), _lib.ptr(ws), 1e-3, 300,
                                               _lib.ptr(mub), _lib.ptr(pb), nb_, st))
            for b in range(B):
                if not (np.array_equal(mub[b], mu0) and np.array_equal(pb[b], p0) and nb_[b] == nit0.value and st[b] == 0):
                    bad += 1
    print("N = %d: 2 x (%d pipelined + 768 batched) fits of %d passes, %.1f s; differing results so far %d" % (
        N, npipe, nit0.value, time.time() - t0, bad), flush=True)
print("total %.1f s, differing results: %d" % (time.time() - t00, bad))
sys.exit(1 if bad else 0)

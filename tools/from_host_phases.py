"""bench.py's extra.from_host_pipelined taken apart: the host's time per call (upload, submit, collect, destroy) over K fits from host
arrays back to back.   python3 tools/from_host_phases.py [K] [window] [low]
window: tables alive at most; low: a collect brings them down to this many (the fits staged meanwhile then share ONE launch: a
context runs four launches at a time, so one fit per launch means four fits in flight)."""
import ctypes
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from frank_amd import _lib as L  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 48
WIN = int(sys.argv[2]) if len(sys.argv) > 2 else 12
LOW = int(sys.argv[3]) if len(sys.argv) > 3 else WIN
EVERY = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # fh_fit_flush behind every so many submissions (0: never -- a collect that meets a staged fit flushes)
f = bench.Fitter(L, 300, 0)
u, v, V, w = mock_disc_visibilities(10_000_000, seed=0, noise_seed=50)
Vre, Vim = np.ascontiguousarray(V.real), np.ascontiguousarray(V.imag)
f.nfit = u.size
L.check(L.lib.fh_ctx_set_range_cache(f.ctx, 0))
pend, alive = [], []
results = []  # (iterations, sha of mu and p) of every fit collected: a table freed too early would show here
draining = [False]
count = [0]
T = {"upload": 0.0, "submit": 0.0, "collect": 0.0, "destroy": 0.0}


def one():
    vis = ctypes.c_void_p()
    t0 = time.perf_counter()
    L.check(L.lib.fh_vis_upload(0, L.ptr(u), L.ptr(v), L.ptr(Vre), L.ptr(Vim), L.ptr(w), w.size, u.size, ctypes.byref(vis)))
    t1 = time.perf_counter()
    alive.append(vis)
    pend.append(f.submit(vis))
    count[0] += 1
    if EVERY and count[0] % EVERY == 0:
        L.check(L.lib.fh_fit_flush(f.ctx))
    t2 = time.perf_counter()
    T["upload"] += t1 - t0
    T["submit"] += t2 - t1
    while len(alive) > (LOW if draining[0] else WIN):
        draining[0] = True
        t3 = time.perf_counter()
        results.append((f.collect(pend.pop(0)), hashlib.sha1(f.mu.tobytes() + f.p.tobytes()).hexdigest()[:12]))
        t4 = time.perf_counter()
        L.lib.fh_vis_destroy(alive.pop(0))
        t5 = time.perf_counter()
        T["collect"] += t4 - t3
        T["destroy"] += t5 - t4
    draining[0] = False


for _ in range(WIN + 4):
    one()
for k in T:
    T[k] = 0.0
t0 = time.perf_counter()
for _ in range(K):
    one()
t_loop = time.perf_counter() - t0
L.check(L.lib.fh_fit_flush(f.ctx))
while pend:
    results.append((f.collect(pend.pop(0)), hashlib.sha1(f.mu.tobytes() + f.p.tobytes()).hexdigest()[:12]))
f.sync()
dt = time.perf_counter() - t0
print("K=%d window=%d low=%d flush every %d: %.1f fits/s over the loop alone (%.2f ms per fit), %.1f with the drain (%.0f ms)" % (K, WIN, LOW, EVERY, K / t_loop, 1e3 * t_loop / K, K / dt, 1e3 * (dt - t_loop)))
print("host ms per fit: " + ", ".join("%s %.2f" % (k, 1e3 * T[k] / K) for k in T))
wg, fb = ctypes.c_int(0), ctypes.c_int64(0)
L.check(L.lib.fh_fit_cluster_info(f.ctx, ctypes.byref(wg), ctypes.byref(fb)))
print("workgroups of the last fit: %d; clusters that did not assemble and were rerun on one unit: %d" % (wg.value, fb.value))
print("results of the %d fits collected: %s" % (len(results), {r: results.count(r) for r in set(results)}))

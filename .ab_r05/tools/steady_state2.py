"""Steady state with C contexts on ONE device sharing the resident table: the binning passes of consecutive fits run on C streams
(a context's binning pass is a serial chain of ten kernels; at steady state that chain is busy 89 % of the time and IS the rate),
each context with its own slots and launch streams.
    python3 tools/steady_state2.py [contexts=2] [steps=3000]       FRANK_AMD_FIT_SLOTS / _STREAMS apply per context
"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from frank_amd import _lib as L  # noqa: E402
from frank_amd.mock import mock_disc_visibilities  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
fs = [bench.Fitter(L, 300, 0) for _ in range(C)]
fs[0].nfit = 10_000_000
fs[0].upload(*mock_disc_visibilities(10_000_000, seed=0, noise_seed=50))
for f in fs[1:]:
    f.nfit, f.vis, f.n = fs[0].nfit, fs[0].vis, fs[0].n
for f in fs:
    f.fit()
slots = L.lib.fh_fit_slots()


def run(k):
    pend = [[] for _ in fs]
    nit = 0
    for i in range(k):
        c = i % C
        f = fs[c]
        if len(pend[c]) == slots:
            nit = f.collect(pend[c].pop(0))
        pend[c].append(f.submit())
    for c, f in enumerate(fs):
        L.check(L.lib.fh_fit_flush(f.ctx))
    for c, f in enumerate(fs):
        for t in pend[c]:
            nit = f.collect(t)
    return nit


run(16)
for f in fs:
    f.sync()
t0 = time.perf_counter()
nit = run(steps)
for f in fs:
    f.sync()
dt = time.perf_counter() - t0
print("%d contexts, %s slots and %s launch streams each: %.0f fits/s (%d steps in %.2f s, %d iterations)" % (
    C, os.environ.get("FRANK_AMD_FIT_SLOTS", "240"), os.environ.get("FRANK_AMD_FIT_STREAMS", "6"), steps / dt, steps, dt, nit), flush=True)

#!/usr/bin/env python3
"""bench.py -- FrankFitter solves/sec on MI355X (BASELINE.json metric).

A "step" is ONE complete fit, end to end on the device, of the configuration the metric is quoted on
(BASELINE.json configs[1]): N = 300 collocation points, 1e7 synthetic mock-disc visibilities already
resident in HBM, Normal GP fit, fp64:
    bin_gram (deproject + J0 design block + Gram)  ->  [RCCL all-reduce in --mode shard]
    ->  scale/unpack M, j  ->  the full power-spectrum iteration to convergence (tol 1e-3).
Every step streams the whole table and runs the whole iteration, and NOTHING is remembered between steps: the timed region takes
turns on a ring of four resident table objects (the same rows) with the context's baseline-range cache off, so every step pays its one look at
(u, v) (the range that sizes the bucket sort, one host round trip -- taken one step ahead on the context's look-ahead stream,
fh_bin_prefetch_range: a pipeline knows its next table) and the histogram + scan of its table
(extra.headline_with_caches is the same region on one table with both kept: what a bootstrap or a sweep that re-bins pays).
A fit slot keeps the band factors of the smoothing matrix T + I for the hyper-parameters it last ran (they depend on
(w_smooth, alpha, p0) and the collocation points only).
`value` = fits completed by all ranks / max-over-ranks wall time.  The timed region holds `steps` binning passes and ONE
drain of the last fits' iterations (~0.09 s), so `value` grows with --steps; extra.steady_state is the rate of a >= 2 s run.

    python bench.py                                  # 1 GPU, defaults finish in a few minutes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: fits are independent objects, so each rank fits its own 1e7-visibility dataset (weak
scaling, no data-path collective): that is `value`.  Whenever WORLD_SIZE > 1 the sharded-visibility path
(BASELINE.json configs[3]: ONE fit of 1e8 visibilities sharded over the ranks, RCCL all-reduce over xGMI of
the packed (N^2+N)-sized sufficient statistics) is timed after the headline region and reported under
"sharded_fit" -- inside a watchdog, so that a collective problem costs that key, never the headline line.
torch is used ONLY for the rendezvous / barrier / max-over-ranks (gloo, CPU tensors); the data path is
libfrank_hip + RCCL.

Whenever WORLD_SIZE > 1 "sweep512_multi" (BASELINE configs[4]: 512 fits of one 1e6-visibility mapping split over the ranks,
the packed statistics handed to every rank by one RCCL all-reduce) runs in the same watchdog.  NOTE: the builder has one
GPU; no N > 1 value has been measured by the builder, the scaling curve is the driver's.

Rank 0 at N=1 also reports, outside the timed region and bounded to about a minute in total:
  extra.steady_state         fits/s of the same pipeline over a >= 2 s run (the drain is < 5 % of it), fit loops resident
  extra.headline_with_caches the timed region of the headline on ONE table, range and histogram caches on (rounds 1-5's `value`)
  extra.steady_state_cached  the steady state on one table with the caches on
  extra.from_host_pipelined  fits from HOST arrays back to back: the upload of fit i + 1 beside the iterations of fit i
  extra.sweep512_distinct    configs[4] with every fit binning its own 1e6-visibility table
  extra.lognormal_fullsize   BASELINE configs[2] (N=300, 1e7 visibilities, LogNormal) on the resident table
  extra.lognormal_N640       the same table at N=640: LogNormal on the persistent kernel's WIDE form (320 < N <= 640, round 6)
  extra.fp32_table           the same table stored in single precision (configs[2]'s "fp32": 20 B per visibility)
  extra.sweep512             BASELINE configs[4] on one GPU (512 fits of one 1e6-visibility mapping)
  extra.uvbin                UVDataBinner streaming passes at 1e7 rows (HBM roofline)
  extra.geometry_fit         FitGeometryFourierBessel (N = 20) of the resident table: seconds per fit, residual pass
  cpu_baseline               the CPU oracle on one core and on all host cores (independent fits per core)
"""
import argparse
import hashlib
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_COLL = 300
N_VIS = 10_000_000
RMAX_ARCSEC = 2.0
HYPER = dict(alpha=1.05, p0=1e-15, wsmooth=1e-4, tol=1e-3, max_iter=2000)
# roofline constants: MI355X fp64 matrix peak (AMD CDNA4 datasheet; the microarch guide lists no fp64 MFMA row)
FP64_MFMA_PEAK_TFLOPS = 78.6
K2_KERNEL_NAME = "fit_loop_kernel"
K1_KERNEL_NAME = "vr_gram_kernel"  # the Gram kernel of the binning pass (on the virtual rows of the buckets)
N_CU = 256
HBM_PEAK_GBPS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # the timed region holds K binning passes back to back plus ONE pipeline drain (the iteration of the last fit,
    # ~0.24 s): the default K amortises it, a small K mostly measures it
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--prewarm-seconds", type=float, default=2.0,
                    help="untimed run of the same pipeline in front of the W warm-up steps: a device that has been idle takes its "
                         "first seconds of load to reach its clocks (the first process on a fresh box measured 328 fits/s at 20 steps, "
                         "the second one 358); 0: none")
    ap.add_argument("--nvis", type=int, default=N_VIS)
    ap.add_argument("--ncoll", type=int, default=N_COLL)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sharded", action="store_true",
                    help="skip the BASELINE configs[3] leg (one fit sharded over the ranks, RCCL all-reduce) that "
                         "otherwise runs whenever WORLD_SIZE > 1")
    ap.add_argument("--sharded-total", type=float, default=1e8, help="visibilities of the sharded fit (all ranks)")
    ap.add_argument("--sharded-cap", type=float, default=2.5e7, help="most visibilities one rank generates / holds")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary workloads (extra.*)")
    ap.add_argument("--force-legs", action="store_true",
                    help="run the two multi-rank legs (sharded_fit, sweep512_multi) even at WORLD_SIZE = 1, over a one-rank "
                         "RCCL communicator: every line of them on a one-GPU box (tests)")
    return ap.parse_args()


class Fitter:
    """Thin ctypes driver of the device-resident path (no host arrays in the timed region)."""

    def __init__(self, L, ncoll, device):
        from frank_amd.constants import rad_to_arcsec
        from frank_amd.mock import MOCK_GEOMETRY
        self.L, self.N = L, ncoll
        self.dht = ctypes.c_void_p()
        L.check(L.lib.fh_dht_create(RMAX_ARCSEC / rad_to_arcsec, ncoll, 0, ctypes.byref(self.dht)))
        self.ctx = ctypes.c_void_p()
        L.check(L.lib.fh_ctx_create(self.dht, device, ctypes.byref(self.ctx)))
        g = MOCK_GEOMETRY
        self.geom = L.fh_geometry(g["inc"], g["PA"], g["dRA"], g["dDec"])
        self.device = device
        self.mu, self.p = np.empty(ncoll), np.empty(ncoll)
        self.niter = ctypes.c_int(0)
        self.vis = None
        self.tables = []
        self.n = 0
        self.nfit = 0

    def upload(self, u, v, V, w):
        L = self.L
        vis = ctypes.c_void_p()
        Vre, Vim = np.ascontiguousarray(V.real), np.ascontiguousarray(V.imag)
        L.check(L.lib.fh_vis_upload(self.device, L.ptr(u), L.ptr(v), L.ptr(Vre), L.ptr(Vim), L.ptr(w), w.size, u.size,
                                    ctypes.byref(vis)))
        self.vis, self.n = vis, u.size
        self.tables.append(vis)
        if not getattr(self, "nfit", 0):
            self.nfit = u.size  # rows of one headline fit (the table may hold more, for the sharded leg)

    def bin(self, count=None, vis=None):
        L = self.L
        L.check(L.lib.fh_bin_reset(self.ctx))
        L.check(L.lib.fh_bin_visibilities(self.ctx, ctypes.byref(self.geom), self.vis if vis is None else vis, 0,
                                          self.nfit if count is None else count))

    def kernel_ms(self):
        ms = ctypes.c_float(0)
        self.L.check(self.L.lib.fh_bin_last_kernel_ms(self.ctx, ctypes.byref(ms)))
        return ms.value

    def prepass_ms(self):
        ms = ctypes.c_float(0)
        self.L.check(self.L.lib.fh_bin_last_prepass_ms(self.ctx, ctypes.byref(ms)))
        return ms.value

    def range_ms(self):
        ms = ctypes.c_float(0)
        self.L.check(self.L.lib.fh_bin_last_range_ms(self.ctx, ctypes.byref(ms)))
        return ms.value

    def loop_kernel_ms(self):
        ms = ctypes.c_float(0)
        self.L.check(self.L.lib.fh_fit_last_kernel_ms(self.ctx, ctypes.byref(ms)))
        return ms.value

    def solve(self):
        L = self.L
        H0, qmn, qmx = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        L.check(L.lib.fh_stats_finalize(self.ctx, ctypes.byref(self.geom), 0, 1, None, None, ctypes.byref(H0),
                                        ctypes.byref(qmn), ctypes.byref(qmx)))
        h = HYPER
        L.check(L.lib.fh_fit_normal(self.ctx, None, None, h["alpha"], h["p0"], h["wsmooth"], h["tol"], h["max_iter"],
                                    L.ptr(self.mu), L.ptr(self.p), ctypes.byref(self.niter), None, None))
        return self.niter.value

    def fit(self):
        self.bin()
        return self.solve()

    def submit(self, vis=None):
        """bin_gram on the main stream, then hand the iteration to a fit slot (fit_loop kernel on its own stream)."""
        L = self.L
        self.bin(vis=vis)
        # nothing is asked back from the finalisation (M, j stay on the device; the baseline range was checked by the
        # synchronous fit of the warm-up): the call does not wait for the binning pass
        L.check(L.lib.fh_stats_finalize(self.ctx, ctypes.byref(self.geom), 0, 0, None, None, None, None, None))
        h = HYPER
        t = ctypes.c_int(-1)
        L.check(L.lib.fh_fit_submit(self.ctx, h["alpha"], h["p0"], h["wsmooth"], h["tol"], h["max_iter"],
                                    ctypes.byref(t)))
        return t.value

    def collect(self, ticket):
        L = self.L
        L.check(L.lib.fh_fit_collect(self.ctx, ticket, L.ptr(self.mu), L.ptr(self.p), ctypes.byref(self.niter)))
        return self.niter.value

    def run_steps(self, k, kernel_ms=None, ring=None):
        """k independent end-to-end fits, pipelined: the iteration of fit i overlaps the binning of fit i+1.
        ring: tables to take turns with (default: the one resident table)."""
        L = self.L
        slots = L.lib.fh_fit_slots()
        pending, nit = [], 0

        def look_ahead(i):
            # the pipeline knows its next table: its one look at (u, v) -- the baseline range that sizes the bucket sort -- runs on
            # the context's look-ahead stream beside the pass in front of it (include/frank_hip.h: fh_bin_prefetch_range)
            if ring:
                L.check(L.lib.fh_bin_prefetch_range(self.ctx, ctypes.byref(self.geom), ring[i % len(ring)], 0, self.nfit))
        look_ahead(0)
        for i in range(k):
            if len(pending) == slots:
                nit = self.collect(pending.pop(0))
            if i + 1 < k:
                look_ahead(i + 1)  # (queued IN FRONT of this step's pass: done long before the host comes back for step i + 1)
            pending.append(self.submit(None if not ring else ring[i % len(ring)]))
            if kernel_ms is not None:
                kernel_ms.append(self.kernel_ms())
        L.check(L.lib.fh_fit_flush(self.ctx))  # the last, partly filled launch
        for t in pending:
            nit = self.collect(t)
        return nit

    def sync(self):
        self.L.check(self.L.lib.fh_ctx_synchronize(self.ctx))

    def cluster_info(self):
        """(workgroups the last synchronous fit ran on, cluster launches of this context that fell back to one CU)"""
        wg, fb = ctypes.c_int(0), ctypes.c_int64(0)
        self.L.check(self.L.lib.fh_fit_cluster_info(self.ctx, ctypes.byref(wg), ctypes.byref(fb)))
        return wg.value, fb.value


def steady_state(f, L, steps=0, ring=0, min_seconds=2.0):
    """fits/s of the pipeline over a run of at least `min_seconds` (the one drain of the last iterations, ~0.1 s, is then
    < 5 % of it).  ring > 0: every step bins a different table of a ring of `ring` resident tables with the range cache of
    the context off -- what a stream of tables the context has never seen costs (one look at (u, v) and one host round trip
    per step more)."""
    from frank_amd.mock import mock_disc_visibilities
    tables = None
    if ring:
        while len(f.tables) < ring:
            k = len(f.tables)
            u, v, V, w = mock_disc_visibilities(f.nfit, seed=7000 + k, noise_seed=7100 + k)
            keep = f.vis
            f.upload(u, v, V, w)
            f.vis = keep
        tables = f.tables[:ring]
        L.check(L.lib.fh_ctx_set_range_cache(f.ctx, 0))
    try:
        k = steps if steps else 400
        # (a pipeline that has been full once: the first window of a context that comes from a drain runs its first fits in the
        #  form of the loop that is faster alone, and the leg that came first among the extras read 3-8 % low for it)
        if not steps:
            f.run_steps(300, ring=tables)
            f.sync()
        while True:
            f.run_steps(8, ring=tables)
            f.sync()
            t0 = time.perf_counter()
            nit = f.run_steps(k, ring=tables)
            f.sync()
            dt = time.perf_counter() - t0
            if dt >= min_seconds or steps:
                break
            k = int(k * max(1.3 * min_seconds / dt, 1.5))
    finally:
        if ring:
            L.check(L.lib.fh_ctx_set_range_cache(f.ctx, 1))
    return {"fits_per_s": k / dt, "steps": k, "seconds": dt, "ms_per_step": 1e3 * dt / k, "fit_slots": L.lib.fh_fit_slots(),
            "iterations_of_the_last_fit": nit,
            "workload": ("the headline step, %d times back to back" % k) if not ring else
                        ("the headline step on a ring of %d resident table objects, baseline-range cache off, range look-ahead" % ring)}


class _stdout_to_stderr:
    """File descriptor 1 pointed at file descriptor 2 for the duration (output of C++ libraries included)."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container on a 256-thread
    host is often given a few cores; oversubscribing them 30x is what a naive os.cpu_count() pool does)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                parts = fh.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                        n = min(n, max(1, q // int(fh.read())))
            break
        except Exception:
            continue
    return n


def _cpu_leg(args):
    """One worker of the CPU baseline: the oracle's map_visibilities on `ns` visibilities and `it` power-spectrum
    iterations, timed separately.  Runs in a fresh interpreter (spawn): no GPU state, imports only numpy + oracle."""
    ncoll, ns, it, seed = args
    from frank_amd.constants import rad_to_arcsec
    from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities
    from oracle import oracle as fo
    g = MOCK_GEOMETRY
    geom = (g["inc"], g["PA"], g["dRA"], g["dDec"])
    u, v, V, w = mock_disc_visibilities(ns, seed=seed, noise_seed=50 + seed)
    t0 = time.perf_counter()
    m = fo.map_visibilities(ncoll, RMAX_ARCSEC / rad_to_arcsec, geom, u, v, V, w)
    t_map = time.perf_counter() - t0
    t0 = time.perf_counter()
    out = fo.frank_fit_normal(ncoll, RMAX_ARCSEC / rad_to_arcsec, m["M"], m["j"], max_iter=it, **{
        k: HYPER[k] for k in ("alpha", "p0", "wsmooth", "tol")})
    t_it = (time.perf_counter() - t0) / max(out["niter"], 1)
    return t_map, t_it, out["niter"]


def cpu_baseline(ncoll, nvis, gpu_niter):
    """The CPU oracle (oracle/frank_oracle.c, a single-threaded port of the reference path) on a bounded sample of the
    same workload: first alone on one core, then one copy on every host core at once (independent fits are the unit of
    work, so a node's CPU throughput is cores / seconds-per-fit with all cores loaded)."""
    ns = min(nvis, 100_000 if ncoll >= 200 else 400_000)
    it = 150
    t_map, t_it, nit = _cpu_leg((ncoll, ns, it, 0))
    t_fit = t_map * (nvis / ns) + t_it * gpu_niter
    out = {"value": 1.0 / t_fit, "unit": "fits/s", "cores": 1, "kind": "port",
           "sample": "oracle map_visibilities on %d of %d visibilities (%.1f s, scaled linearly) + %d of the %d "
                     "power-spectrum iterations (%.1f ms/iteration, scaled)" % (ns, nvis, t_map, nit, gpu_niter, 1e3 * t_it),
           "s_per_fit": t_fit,
           "reference_s_per_fit": 196.0,
           "reference_note": "the reference itself (NumPy/SciPy + BLAS, 1 thread) measured in the build container: "
                             "196 s/fit (BASELINE.md); the naive-loop C port is slower than that, quote 196 s"}
    try:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        cores = usable_cpus()
        workers = max(1, min(cores, 64))  # bounded: the sample is per core, more copies add nothing
        with ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn")) as pool:
            res = list(pool.map(_cpu_leg, [(ncoll, ns, it, 1 + k) for k in range(workers)]))
        tm = float(np.median([r[0] for r in res]))
        ti = float(np.median([r[1] for r in res]))
        t_loaded = tm * (nvis / ns) + ti * gpu_niter
        out["all_cores"] = {"value": workers / t_loaded, "unit": "fits/s", "cores": workers, "usable_cpus": cores, "host_cpus": os.cpu_count(),
                            "s_per_fit_per_core_loaded": t_loaded,
                            "sample": "%d concurrent single-thread copies of the sample above (median %.1f s map, "
                                      "%.1f ms/iteration), one independent fit per core" % (workers, tm, 1e3 * ti)}
    except Exception as e:  # the one-core figure stands on its own
        out["all_cores"] = {"error": repr(e)}
    return out


def extras(f, L, a):
    """Secondary workloads on rank 0 at N=1, after the timed region (bounded: ~16 s LogNormal + ~2 s sweep + ~2 s)."""
    ex = {}
    N = a.ncoll
    h = HYPER
    H0, qmn, qmx = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    # -- what the pipeline of the headline does at steady state (>= 2 s runs), and on tables it has never seen
    try:
        # the timed region of rounds 1-5: the same steps on ONE table, baseline-range and histogram caches on
        f.run_steps(a.warmup)
        f.sync()
        t0 = time.perf_counter()
        f.run_steps(a.steps)
        f.sync()
        dtc = time.perf_counter() - t0
        ex["headline_with_caches"] = {"fits_per_s": a.steps / dtc, "steps": a.steps, "ms_per_step": 1e3 * dtc / a.steps,
                                      "workload": "the timed region of the headline on one resident table: the context keeps the "
                                                  "baseline range and the (u, v) histogram + scan of the rows it binned last"}
    except Exception as e:
        ex["headline_with_caches"] = {"error": repr(e)}
    try:
        ex["steady_state"] = steady_state(f, L, ring=4)
        ex["steady_state_cached"] = steady_state(f, L)
        # ... and with the fit loops in their register-resident form (fit_loop_rr.hip, FRANK_AMD_K2_RR=1, read at every launch;
        # same bits): opt-in -- a loop ALONE takes 172 us per pass in that form against 135, with the device full 183 against
        # 196 --, so the drained runs above and the sweeps keep the forms that work in memory
        prev_rr = os.environ.get("FRANK_AMD_K2_RR")
        os.environ["FRANK_AMD_K2_RR"] = "1"
        try:
            ex["steady_state_register_resident"] = steady_state(f, L, ring=4)
            ex["steady_state_register_resident"]["workload"] += "; fit loops with the matrix resident in registers (FRANK_AMD_K2_RR=1)"
        finally:
            if prev_rr is None:
                del os.environ["FRANK_AMD_K2_RR"]
            else:
                os.environ["FRANK_AMD_K2_RR"] = prev_rr
        for t in f.tables[1:]:
            L.lib.fh_vis_destroy(t)
        del f.tables[1:]
    except Exception as e:
        ex.setdefault("steady_state", {})["error"] = repr(e)
    # -- the fit loop with the DEVICE FULL: 256 identical fits of the headline mapping resident in one launch, one compute unit
    #    each (fh_fit_normal_batched), with the clock probe on -- the form a deep sweep runs, and the roofline of the loaded kernel
    #    (a loaded pass is bound by the bytes it moves beyond the L2: profiles/r05_pmc_fit_loop_loaded.json)
    prev = os.environ.get("FRANK_AMD_SWEEP_NO_CLUSTERS")
    os.environ["FRANK_AMD_SWEEP_NO_CLUSTERS"] = "1"
    try:
        Bf = 256
        o3 = (ctypes.c_int64 * 3)()
        f.bin()
        L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 0, None, None, None, None, None))
        al, p0v, wsv = np.full(Bf, h["alpha"]), np.full(Bf, h["p0"]), np.full(Bf, h["wsmooth"])
        mu_b, p_b = np.empty((Bf, N)), np.empty((Bf, N))
        nit_b, st_b = (ctypes.c_int * Bf)(), (ctypes.c_int * Bf)()
        def resident_launch():
            best = None
            for _ in range(2):
                L.check(L.lib.fh_ctx_loop_clocks(f.ctx, 1, o3))
                f.sync()
                t0 = time.perf_counter()
                L.check(L.lib.fh_fit_normal_batched(f.ctx, None, None, Bf, L.ptr(al), L.ptr(p0v), L.ptr(wsv), h["tol"], h["max_iter"],
                                                    L.ptr(mu_b), L.ptr(p_b), nit_b, st_b))
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            L.check(L.lib.fh_ctx_loop_clocks(f.ctx, 0, o3))
            return best
        prev_rr = os.environ.get("FRANK_AMD_K2_RR")
        os.environ["FRANK_AMD_K2_RR"] = "1"
        try:
            best_rr = resident_launch()
            rr = {"fits_per_s": Bf / best_rr, "s_total": best_rr, "us_per_pass_on_the_device": o3[1] / 100.0 / max(o3[2], 1),
                  "sha256_mu_p": hashlib.sha256(mu_b.tobytes() + p_b.tobytes()).hexdigest()}
        finally:
            if prev_rr is None:
                del os.environ["FRANK_AMD_K2_RR"]
            else:
                os.environ["FRANK_AMD_K2_RR"] = prev_rr
        best = resident_launch()
        rr["same_bits_as_the_default_form"] = rr.pop("sha256_mu_p") == hashlib.sha256(mu_b.tobytes() + p_b.tobytes()).hexdigest()
        passes = nit_b[0] + 2
        tf = Bf / best * passes * (2.0 * (N + 1) ** 3 / 3.0) / 1e12
        ex["device_full"] = {"workload": "%d identical N=%d fits resident in ONE launch, one compute unit each (the batched form of "
                                         "the sweeps; the form launch_loop takes for a launch that fills the device: the matrix "
                                         "resident in the vector registers of its compute unit, fit_loop_rr.hip)" % (Bf, N),
                             "fits_per_s": Bf / best, "s_total": best, "passes_per_fit": passes,
                             "us_per_pass_on_the_device": o3[1] / 100.0 / max(o3[2], 1), "clock_MHz": 100.0 * o3[0] / max(o3[1], 1),
                             "roofline": {"bound": "mfma", "achieved": tf, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                          "frac": tf / FP64_MFMA_PEAK_TFLOPS,
                                          "note": "algorithmic flops (2 n^3 / 3 per pass) of all fits over the wall time of the launch, "
                                                  "against the whole chip's fp64 matrix peak; what holds it there is instruction issue "
                                                  "(a wave issues a matrix instruction per 64 cycles at most, and with two waves on a SIMD "
                                                  "a vector instruction costs ~65 cycles while the other's matrix instruction runs: "
                                                  "profiles/r05_mfma_f64_issue.txt), not memory: 0.23 MB per pass beyond the L2 against the "
                                                  "5.0 MB of the forms that work in memory (profiles/r05_pmc_fit_loop_256_resident*.json)"},
                             "forced_register_resident_form": rr}
    except Exception as e:
        ex["device_full"] = {"error": repr(e)}
    finally:
        if prev is None:
            del os.environ["FRANK_AMD_SWEEP_NO_CLUSTERS"]
        else:
            os.environ["FRANK_AMD_SWEEP_NO_CLUSTERS"] = prev

    def finalize():
        L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 1, None, None, ctypes.byref(H0),
                                        ctypes.byref(qmn), ctypes.byref(qmx)))
    # -- BASELINE configs[1] as a user runs it (fit.py:455-471 hands NumPy arrays): FrankFitter.fit(u, v, V, w) from HOST arrays --
    #    upload over PCIe + binning pass + fit --, and the upload alone from pageable and from pinned host memory
    try:
        from frank_amd import FixedGeometry, FrankFitter
        from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities
        u, v, V, w = mock_disc_visibilities(f.nfit, seed=0, noise_seed=50)
        FF = FrankFitter(RMAX_ARCSEC, N, FixedGeometry(**MOCK_GEOMETRY), alpha=h["alpha"], weights_smooth=h["wsmooth"],
                         tol=h["tol"], max_iter=h["max_iter"], verbose=False, store_iteration_diagnostics=True)
        FF.fit(u, v, V, w)
        nit_h = int(FF.iteration_diagnostics["num_iterations"])
        # (timed on a fitter that does not keep per-iteration diagnostics -- 3 MB of copies and a GEMV per pass --: the first call
        #  of a fresh fitter pays its context's workspaces and first look at the table, the later ones are what a loop over
        #  tables pays per fit)
        FF2 = FrankFitter(RMAX_ARCSEC, N, FixedGeometry(**MOCK_GEOMETRY), alpha=h["alpha"], weights_smooth=h["wsmooth"],
                          tol=h["tol"], max_iter=h["max_iter"], verbose=False)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            FF2.fit(u, v, V, w)
            ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); m = FF2.preprocess_visibilities(u, v, V, w); t_map = time.perf_counter() - t0
        t0 = time.perf_counter(); FF2.fit_preprocessed(m); t_fit = time.perf_counter() - t0
        Vre, Vim = np.ascontiguousarray(V.real), np.ascontiguousarray(V.imag)
        nbytes = 8.0 * (u.size * 4 + w.size)

        def upload(arrs):
            best = None
            for _ in range(3):
                vis = ctypes.c_void_p()
                t0 = time.perf_counter()
                L.check(L.lib.fh_vis_upload(f.device, L.ptr(arrs[0]), L.ptr(arrs[1]), L.ptr(arrs[2]), L.ptr(arrs[3]), L.ptr(arrs[4]),
                                            arrs[4].size, arrs[0].size, ctypes.byref(vis)))
                dt = time.perf_counter() - t0
                L.lib.fh_vis_destroy(vis)
                best = dt if best is None else min(best, dt)
            return best
        t_page = upload((u, v, Vre, Vim, w))
        pinned = None
        try:  # the same five columns in pinned host memory (hipHostMalloc through the HIP runtime the library already loaded)
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
            bufs, arrs = [], []
            for x in (u, v, Vre, Vim, w):
                ptr = ctypes.c_void_p()
                if hip.hipHostMalloc(ctypes.byref(ptr), x.nbytes, 0) != 0:
                    raise RuntimeError("hipHostMalloc failed")
                bufs.append(ptr)
                a_ = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_double)), shape=(x.size,))
                a_[:] = x
                arrs.append(a_)
            pinned = upload(arrs)
            del arrs
            for ptr in bufs:
                hip.hipHostFree(ptr)
        except Exception as e:  # noqa: BLE001
            pinned = repr(e)
        ex["from_host_arrays"] = {
            "workload": "FrankFitter(Rmax=%g, N=%d).fit(u, v, V, w) from NumPy arrays, %d visibilities (BASELINE configs[1] as "
                        "fit.py:455-471 runs it): upload + binning pass + fit" % (RMAX_ARCSEC, N, u.size),
            "s_per_fit": min(ts), "s_first_call": ts[0], "iterations": nit_h,
            "preprocess_visibilities_s": t_map, "fit_preprocessed_s": t_fit,
            "upload_bytes": nbytes, "upload_pageable_s": t_page, "upload_pageable_GBps": nbytes / t_page / 1e9,
            "upload_pinned_s": pinned if isinstance(pinned, float) else None,
            "upload_pinned_GBps": (nbytes / pinned / 1e9) if isinstance(pinned, float) else pinned,
            "note": "the headline `value` starts with the table resident in HBM; this is the same fit paying PCIe (the complex "
                    "visibilities go up as NumPy holds them and are split into two columns on the device)"}
        # ... and fits from host arrays BACK TO BACK through the pipeline: the upload of table i + 1 (a blocking copy of the host
        # thread) runs beside the iterations of the fits before it (their loops sit on the launch streams): the rate is what
        # PCIe and the host-side split of the complex column allow, not upload + fit one behind the other
        try:
            K = 96
            pend, alive = [], []
            count = [0]

            def one():
                vis = ctypes.c_void_p()
                L.check(L.lib.fh_vis_upload(f.device, L.ptr(u), L.ptr(v), L.ptr(Vre), L.ptr(Vim), L.ptr(w), w.size, u.size, ctypes.byref(vis)))
                alive.append(vis)
                pend.append(f.submit(vis))
                count[0] += 1
                # (a context runs four launches at a time and a collect that meets a fit still staged launches it and waits a whole
                #  fit for it: the staged fits go out every eight submissions -- launches of eight fits on clusters, 45 ms each, one
                #  every 60 ms -- so that the fit a collect asks for, submitted 24 uploads ago, has long ended.  tools/from_host_phases.py)
                if count[0] % 8 == 0:
                    L.check(L.lib.fh_fit_flush(f.ctx))
                while len(alive) > 24:  # (a table is freed behind the fit that binned it -- fh_vis_destroy waits for the streams that
                                        #  read tables, not for the fit loops in flight --; 24 tables = 9.6 GB)
                    f.collect(pend.pop(0))
                    L.lib.fh_vis_destroy(alive.pop(0))
            L.check(L.lib.fh_ctx_set_range_cache(f.ctx, 0))
            for _ in range(32):  # (the pipeline full when the clock starts)
                one()
            t0 = time.perf_counter()
            for _ in range(K):
                one()
            L.check(L.lib.fh_fit_flush(f.ctx))
            while pend:
                f.collect(pend.pop(0))
            f.sync()
            dtp = time.perf_counter() - t0
            for vis in alive:
                L.lib.fh_vis_destroy(vis)
            ex["from_host_pipelined"] = {
                "workload": "%d fits of %d visibilities each from HOST arrays (five fp64 columns, pageable), back to back: upload of "
                            "table i + 1 beside the iterations of the fits before it; range cache off" % (K, u.size),
                "fits_per_s": K / dtp, "s_total": dtp, "ms_per_fit": 1e3 * dtp / K,
                "upload_bound_fits_per_s": 1.0 / t_page,
                "note": "bound by the upload (%.1f ms per table at %.0f GB/s): the copy is the host thread's, the device iterates "
                        "meanwhile" % (1e3 * t_page, nbytes / t_page / 1e9)}
        except Exception as e:  # noqa: BLE001
            ex["from_host_pipelined"] = {"error": repr(e)}
        finally:
            L.check(L.lib.fh_ctx_set_range_cache(f.ctx, 1))
        del u, v, V, w, Vre, Vim
    except Exception as e:  # noqa: BLE001
        ex["from_host_arrays"] = {"error": repr(e)}
    # -- BASELINE configs[2]: LogNormal fit (alpha = 1.3, w_smooth = 1e-2 as the reference's own LogNormal test,
    #    tests.py:350) of the resident table, end to end; with the default line search (S^-1 (x + lam p) by linearity) and
    #    with the reference's arithmetic (every trial point multiplied out) -- include/frank_hip.h
    def lognormal_roofline(n, iters, steps, nfev, nhess, seconds, workgroups=8):
        """Algorithmic flops of a LogNormal fit from its own counters against the fp64 matrix peak of the compute
        units the fit holds (a cluster of eight workgroups of one XCD from N = 160 on)."""
        fl = {"S^-1 = Y^T diag(1/p) Y, symmetric half (per iteration)": iters * 1.0 * n ** 3,
              "Tr2: triangular solve with n right-hand sides (per iteration)": iters * 1.0 * n ** 3,
              "Cholesky of the Hessian (per Newton Hessian and per iteration, for Tr2)": (nhess + iters) * n ** 3 / 3.0,
              "Newton directions: two triangular solves (per step)": steps * 2.0 * n ** 2,
              "objective: M I (per evaluation) and S^-1 p (per step)": (nfev + steps) * 2.0 * n ** 2,
              "Hessian builds": (nhess + iters) * 3.0 * n ** 2}
        total = float(sum(fl.values()))
        peak = FP64_MFMA_PEAK_TFLOPS * workgroups / N_CU
        traffic, src = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "r06_pmc_lognormal.json")) as fh:
                d = json.load(fh)
            e = [v for k, v in d.items() if k.startswith("lognormal_kernel")][0]
            traffic = int(e["hbm_bytes_per_launch"])
            src = "static: profiles/r06_pmc_lognormal.json (rocprofv3 --pmc on tools/ln_fullsize.py), library build '%s'" % d.get("_library")
        except Exception:  # noqa: BLE001
            pass
        return {"kernel": "lognormal_kernel (cluster of %d workgroups)" % workgroups, "bound": "mfma", "achieved": total / seconds / 1e12,
                "peak": peak, "unit": "TFLOP/s", "frac": total / seconds / 1e12 / peak, "algorithmic_flops": total,
                "flops_by_phase": fl, "traffic": traffic, "traffic_source": src,
                "note": "the matrix peak is the nominal roof (the products run on v_mfma_f64_16x16x4_f64); what holds the kernel is "
                        "what runs on ONE compute unit: a Newton direction is two substitutions on 0.72 MB of factors (~25 B per cycle "
                        "from the L2, a 64-step chain per 64-row block), an evaluation 0.72 MB of M; the Hessian's factorisation keeps "
                        "its trailing tiles in the registers of the cluster's helpers since the second half of round 6 (366 k -> 254 k "
                        "cycles: the chain of diagonal tiles and the panel behind it are what a step waits for) -- "
                        "profiles/r06_lognormal_phases.txt, r06_lognormal_dist_cholesky.txt"}

    def lognormal(reference_products):
        s_map, p = np.empty(N), np.empty(N)
        nit = ctypes.c_int(0)
        stats = (ctypes.c_int64 * 9)()
        L.check(L.lib.fh_ctx_set_lognormal_linesearch(f.ctx, reference_products))
        if not reference_products:
            # (the first launch of the LogNormal kernel in a process can pay a one-time ~0.2 s -- the runtime growing its scratch
            #  memory behind a drained queue; seen with and without the distributed Cholesky, tools/ln_after_pipeline.py --: one
            #  two-pass fit, untimed, as the headline has its warm-up steps)
            f.bin()
            finalize()
            L.check(L.lib.fh_fit_lognormal(f.ctx, None, None, 1.3, 1e-35, 1e-2, h["tol"], 2, 1e5, L.ptr(s_map), L.ptr(p),
                                           ctypes.byref(nit), None, stats, None, None))
        t0 = time.perf_counter()
        f.bin()
        finalize()
        t1 = time.perf_counter()
        L.check(L.lib.fh_fit_lognormal(f.ctx, None, None, 1.3, 1e-35, 1e-2, h["tol"], h["max_iter"], 1e5, L.ptr(s_map),
                                       L.ptr(p), ctypes.byref(nit), None, stats, None, None))
        t2 = time.perf_counter()
        L.check(L.lib.fh_ctx_set_lognormal_linesearch(f.ctx, 0))
        I = np.exp(s_map + np.log(1e5))
        return {"workload": "BASELINE configs[2]: N=%d, %d visibilities, LogNormal, alpha=1.3, w_smooth=1e-2, fp64 "
                            "arithmetic" % (N, f.nfit),
                "linesearch": "reference" if reference_products else "linear",
                "s_per_fit": t2 - t0, "bin_s": t1 - t0, "fit_s": t2 - t1,
                "power_spectrum_iterations": nit.value, "newton_steps": int(stats[1]),
                "function_evaluations": int(stats[2]), "hessian_factorisations": int(stats[3]),
                "ms_per_power_spectrum_iteration": 1e3 * (t2 - t1) / max(nit.value, 1),
                "I_min": float(I.min()), "I_max": float(I.max()), "finite": bool(np.all(np.isfinite(I))),
                "roofline": lognormal_roofline(N, nit.value, int(stats[1]), int(stats[2]), int(stats[3]), t2 - t1)}, I
    try:
        ex["lognormal_fullsize"], I_lin = lognormal(0)
        ex["lognormal_fullsize_reference_linesearch"], I_ref = lognormal(1)
        ex["lognormal_fullsize_reference_linesearch"]["note"] = (
            "the reference's Newton solves end in round-off (it ignores their exit status): the number of steps and "
            "Hessians, and with them the seconds, depend on 1e-16-level changes of M")
        ex["lognormal_fullsize"]["profile_vs_reference_linesearch_max_abs_diff_over_max"] = float(
            np.abs(I_lin - I_ref).max() / np.abs(I_ref).max())
        ex["lognormal_fullsize"]["note"] = (
            "the DEFAULT line search forms S^-1 (x + lam p) by linearity: not the reference's arithmetic (held to the reference's whole "
            "fit at 1e-4 of the maximum, tests/test_gpu_configs.py); the mode that multiplies every trial point out as the reference "
            "does is lognormal_fullsize_reference_linesearch -- quote both")
    except Exception as e:
        ex.setdefault("lognormal_fullsize", {})["error"] = repr(e)
    # -- method='LogNormal' at N = 640 (the persistent kernel's WIDE form, 320 < N <= 640; rounds 4-5: the host-driven route,
    #    lognormal_wide.hip, which FRANK_AMD_LN_WIDE=host still selects) on the same resident table, at most 200 passes of the
    #    power-spectrum loop
    try:
        f2 = Fitter(L, 640, f.device)
        f2.vis, f2.nfit, f2.n = f.vis, f.nfit, f.n
        s2, p2 = np.empty(640), np.empty(640)
        nit2 = ctypes.c_int(0)
        st2 = (ctypes.c_int64 * 9)()
        H0, qmn, qmx = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        t0 = time.perf_counter()
        f2.bin()
        L.check(L.lib.fh_stats_finalize(f2.ctx, ctypes.byref(f2.geom), 0, 1, None, None, ctypes.byref(H0), ctypes.byref(qmn),
                                        ctypes.byref(qmx)))
        t1 = time.perf_counter()
        L.check(L.lib.fh_fit_lognormal(f2.ctx, None, None, 1.3, 1e-35, 1e-2, h["tol"], 200, 1e5, L.ptr(s2), L.ptr(p2),
                                       ctypes.byref(nit2), None, st2, None, None))
        t2 = time.perf_counter()
        I2 = np.exp(s2 + np.log(1e5))
        ex["lognormal_N640"] = {
            "workload": "N=640, %d visibilities, LogNormal, alpha=1.3, w_smooth=1e-2, max_iter=200: the persistent kernel in its "
                        "WIDE form (320 < N <= 640: one LDS panel for the Cholesky laid over the work vectors, round 6; "
                        "FRANK_AMD_LN_WIDE=host keeps round 4's host-driven MinimizeNewton: 39 ms per iteration)" % f.nfit,
            "s_per_fit": t2 - t0, "bin_s": t1 - t0, "fit_s": t2 - t1, "power_spectrum_iterations": nit2.value,
            "newton_steps": int(st2[1]), "function_evaluations": int(st2[2]), "hessian_factorisations": int(st2[3]),
            "ms_per_power_spectrum_iteration": 1e3 * (t2 - t1) / max(nit2.value, 1),
            "I_max": float(I2.max()), "finite": bool(np.all(np.isfinite(I2)))}
        del f2
    except Exception as e:  # noqa: BLE001
        ex["lognormal_N640"] = {"error": repr(e)}
    # -- the binning pass when the baselines REACH the last collocation frequency -- how frank is normally set up (N chosen so
    #    that q[-1] just clears the data, statistical_models.py:512-535): the same rows stretched to 0.95 Q_max fall into ~1 900
    #    buckets of J0 arguments (55 MB of Taylor tables) instead of the 243 of the headline table (13 % of Q_max)
    try:
        from frank_amd.mock import mock_disc_visibilities
        from frank_amd.constants import rad_to_arcsec
        u, v, V, w = mock_disc_visibilities(f.nfit, seed=0, noise_seed=50)
        from frank_amd import FixedGeometry
        from frank_amd.mock import MOCK_GEOMETRY
        ud, vd = FixedGeometry(**MOCK_GEOMETRY).deproject(u, v)
        qmax_data = float(np.hypot(ud, vd).max())
        del ud, vd
        dht_q = np.empty(N)
        L.check(L.lib.fh_dht_get(f.dht, None, L.ptr(dht_q), None, None, None, None, None))
        stretch = 0.95 * dht_q[-1] / qmax_data
        u *= stretch
        v *= stretch
        keep, keepn = f.vis, f.n
        f.upload(u, v, V, w)
        wide = f.tables.pop()
        f.vis, f.n = keep, keepn
        try:
            ms = []
            for _ in range(5):
                f.bin(vis=wide)
                f.sync()
                ms.append(f.prepass_ms() + f.kernel_ms())
            Mw, jw = np.empty((N, N)), np.empty(N)
            L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 1, L.ptr(Mw), L.ptr(jw), ctypes.byref(H0),
                                            ctypes.byref(qmn), ctypes.byref(qmx)))
            # parity: the rows themselves on the matrix pipe (FRANK_AMD_K1=rows, a second context), 1e6 of the rows
            nchk = min(f.nfit, 1_000_000)
            f.bin(nchk, vis=wide)
            Mm, jm = np.empty((N, N)), np.empty(N)
            L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 0, L.ptr(Mm), L.ptr(jm), ctypes.byref(H0),
                                            ctypes.byref(qmn), ctypes.byref(qmx)))
            os.environ["FRANK_AMD_K1"] = "rows"
            try:
                ctx2 = ctypes.c_void_p()
                L.check(L.lib.fh_ctx_create(f.dht, f.device, ctypes.byref(ctx2)))
            finally:
                del os.environ["FRANK_AMD_K1"]
            try:
                L.check(L.lib.fh_bin_reset(ctx2))
                L.check(L.lib.fh_bin_visibilities(ctx2, ctypes.byref(f.geom), wide, 0, nchk))
                Mr, jr = np.empty((N, N)), np.empty(N)
                L.check(L.lib.fh_stats_finalize(ctx2, ctypes.byref(f.geom), 0, 0, L.ptr(Mr), L.ptr(jr), ctypes.byref(H0),
                                                ctypes.byref(qmn), ctypes.byref(qmx)))
            finally:
                L.lib.fh_ctx_destroy(ctx2)
            pm = float(np.median(ms[2:]))
            ex["wide_uv"] = {"workload": "the headline rows stretched by %.2f: baselines to 0.95 x the last collocation frequency "
                                         "(q_max = %.3e of q[-1] = %.3e)" % (stretch, 0.95 * dht_q[-1], dht_q[-1]),
                             "binning_pass_ms": pm, "first_pass_ms": ms[0],
                             "GBps_of_40B_per_vis": 40.0 * f.nfit / (pm * 1e-3) / 1e9,
                             "frac_of_hbm_peak": 40.0 * f.nfit / (pm * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                             "rows_checked_against_the_rows_path": nchk,
                             "M_max_rel_diff_vs_rows_path": float(np.abs(Mm - Mr).max() / np.abs(Mr).max()),
                             "j_max_rel_diff_vs_rows_path": float(np.abs(jm - jr).max() / np.abs(jr).max()),
                             "finite": bool(np.all(np.isfinite(Mw)) and np.all(np.isfinite(jw)))}
        finally:
            L.lib.fh_vis_destroy(wide)
        del u, v, V, w
    except Exception as e:  # noqa: BLE001
        ex["wide_uv"] = {"error": repr(e)}
    # -- "fp32" of BASELINE configs[2] = single-precision STORAGE: the table handed over as float32 / complex64 (20 B per
    #    visibility, fh_vis_upload_f32) is widened as the pre-pass reads it and binned by the same fp64 moments pass.
    #    (Single-precision ARITHMETIC of the design block exists for tables up to 2e6 rows -- fh_ctx_set_arithmetic,
    #    8.8 ms per 1e7 rows and a Gram that is no longer positive definite at that size: measured in round 2, retired here)
    try:
        from frank_amd.mock import mock_disc_visibilities
        u, v, V, w = mock_disc_visibilities(f.nfit, seed=0, noise_seed=50)
        f4 = [np.ascontiguousarray(x, dtype=np.float32) for x in (u, v, V.real, V.imag, w)]
        del u, v, V, w
        vis32 = ctypes.c_void_p()
        L.check(L.lib.fh_vis_upload_f32(f.device, L.fptr(f4[0]), L.fptr(f4[1]), L.fptr(f4[2]), L.fptr(f4[3]), L.fptr(f4[4]),
                                        f4[4].size, f4[0].size, ctypes.byref(vis32)))
        del f4
        try:
            ms = []
            for _ in range(4):
                f.bin(vis=vis32)
                f.sync()
                ms.append(f.prepass_ms() + f.kernel_ms())
            finalize()
            mu32, p32, n32 = np.empty(N), np.empty(N), ctypes.c_int(0)
            L.check(L.lib.fh_fit_normal(f.ctx, None, None, h["alpha"], h["p0"], h["wsmooth"], h["tol"], h["max_iter"],
                                        L.ptr(mu32), L.ptr(p32), ctypes.byref(n32), None, None))
            f.bin()
            finalize()
            mu64, p64, n64 = np.empty(N), np.empty(N), ctypes.c_int(0)
            L.check(L.lib.fh_fit_normal(f.ctx, None, None, h["alpha"], h["p0"], h["wsmooth"], h["tol"], h["max_iter"],
                                        L.ptr(mu64), L.ptr(p64), ctypes.byref(n64), None, None))
            pm = float(np.median(ms[1:]))
            ex["fp32_table"] = {"workload": "the headline table stored in single precision (20 B per visibility), fp64 arithmetic",
                                "binning_pass_ms": pm, "GBps_of_20B_per_vis": 20.0 * f.nfit / (pm * 1e-3) / 1e9,
                                "iterations_fp32_table_vs_fp64_table": [n32.value, n64.value],
                                "profile_max_abs_diff_over_max": float(np.abs(mu32 - mu64).max() / np.abs(mu64).max()),
                                "north_star_tolerance_fp32": 1e-3}
        finally:
            L.lib.fh_vis_destroy(vis32)
    except Exception as e:
        ex["fp32_table"] = {"error": repr(e)}
    # -- BASELINE configs[4] on one GPU: 512 fits (32 alpha x 16 w_smooth) of ONE mapping of 1e6 visibilities
    try:
        al, ws = np.meshgrid(np.linspace(1.01, 1.5, 32), np.logspace(-4, -1, 16))
        al, ws = np.ascontiguousarray(al.ravel()), np.ascontiguousarray(ws.ravel())
        B = al.size
        p0 = np.full(B, h["p0"])
        mu, pp = np.empty((B, N)), np.empty((B, N))
        niter = (ctypes.c_int * B)()
        status = (ctypes.c_int * B)()
        nv = min(f.n, 1_000_000)
        dt = None
        for _ in range(2):  # (the first sweep of a process pays ~50 ms of allocations and code loading; the second is what a user's
                            #  next sweep costs: the better of the two is reported, both are in `s_total_both`)
            t0 = time.perf_counter()
            f.bin(nv)
            finalize()
            L.check(L.lib.fh_fit_normal_batched(f.ctx, None, None, B, L.ptr(al), L.ptr(p0), L.ptr(ws), h["tol"],
                                                h["max_iter"], L.ptr(mu), L.ptr(pp), niter, status))
            d1 = time.perf_counter() - t0
            both = [d1] if dt is None else both + [d1]
            dt = d1 if dt is None else min(dt, d1)
        its = np.array(list(niter))
        ex["sweep512"] = {"workload": "BASELINE configs[4] on ONE GPU: %d fits (alpha x w_smooth grid), N=%d, %d "
                                      "visibilities, shared (M, j) as fit.py:534-548" % (B, N, nv),
                          "fits_per_s": B / dt, "s_total": dt, "s_total_both": both,
                          "schedule": ("single launch, no clusters (FRANK_AMD_SWEEP_NO_CLUSTERS)" if os.environ.get("FRANK_AMD_SWEEP_NO_CLUSTERS") else
                                       "staged (capi_fit.hip: sweep_staged): every fit on one compute unit in a launch that fills the "
                                       "device until %s; the fits still running then continue on clusters of workgroups"
                                       % (("it has made %s passes" % os.environ["FRANK_AMD_SWEEP_CAP"]) if int(os.environ.get("FRANK_AMD_SWEEP_CAP", "0")) > 0
                                          else "every fit has been handed out and 32 are left (checked every 16 passes)")),
                          "iterations_min_median_max": [int(its.min()), int(np.median(its)), int(its.max())],
                          "failed": int(np.sum(np.array(list(status)) != 0)),
                          "not_converged": int(np.sum(its >= h["max_iter"]))}
        # ranking the sweep: marginal likelihood, log prior and Laplace evidence of all points, batched on the device
        # (fh_sweep_evidence; the reference: dense O(N^3) host algebra per point, radial_fitters.py:951-967)
        try:
            sll, lpr, lev = np.empty(B), np.empty(B), np.empty(B)
            t0 = time.perf_counter()
            L.check(L.lib.fh_sweep_evidence(f.ctx, None, None, float(H0.value), B, L.ptr(pp), L.ptr(mu), L.ptr(al), L.ptr(p0),
                                            L.ptr(ws), L.ptr(sll), L.ptr(lpr), L.ptr(lev), None))
            dte = time.perf_counter() - t0
            ok = np.isfinite(lev)
            best = int(np.nanargmax(lev))
            ex["sweep512"]["evidence"] = {"s_total": dte, "points_per_s": B / dte, "finite": int(ok.sum()),
                                          "best_point": {"alpha": float(al[best]), "w_smooth": float(ws[best]),
                                                         "log_evidence": float(lev[best])}}
        except Exception as e:  # noqa: BLE001
            ex["sweep512"]["evidence"] = {"error": repr(e)}
    except Exception as e:
        ex["sweep512"] = {"error": repr(e)}
    # -- BASELINE configs[4], the distinct-datasets variant SURVEY 8(d) asks for beside the shared-(M, j) sweep: every one of the 512
    #    fits bins its OWN 1e6-visibility table (eight resident tables, seeds 0 .. 7, taken in turn: generating 512 on the host would
    #    take minutes and the device work is the same) and runs its grid point, through the pipeline
    try:
        NT, NV = 8, 1_000_000
        f5 = Fitter(L, N, f.device)
        f5.nfit = NV
        tabs = []
        for sd in range(NT):
            f5.upload(*mock_disc_visibilities(NV, seed=sd, noise_seed=50 + sd))
            tabs.append(f5.vis)
        L.check(L.lib.fh_ctx_set_range_cache(f5.ctx, 0))
        grid = [(float(x), float(y)) for x in np.linspace(1.01, 1.5, 32) for y in np.logspace(-4, -1, 16)]
        slots = L.lib.fh_fit_slots()

        def run_grid(points):
            pend, its = [], []
            for i, (ga, gw) in enumerate(points):
                if len(pend) == slots:
                    its.append(f5.collect(pend.pop(0)))
                if i + 1 < len(points):
                    L.check(L.lib.fh_bin_prefetch_range(f5.ctx, ctypes.byref(f5.geom), tabs[(i + 1) % NT], 0, NV))
                f5.bin(vis=tabs[i % NT])
                L.check(L.lib.fh_stats_finalize(f5.ctx, ctypes.byref(f5.geom), 0, 0, None, None, None, None, None))
                t = ctypes.c_int(-1)
                L.check(L.lib.fh_fit_submit(f5.ctx, ga, h["p0"], gw, h["tol"], h["max_iter"], ctypes.byref(t)))
                pend.append(t.value)
            L.check(L.lib.fh_fit_flush(f5.ctx))
            its += [f5.collect(t) for t in pend]
            return its
        run_grid(grid[:32])
        f5.sync()
        t0 = time.perf_counter()
        its5 = run_grid(grid)
        f5.sync()
        dt5 = time.perf_counter() - t0
        ex["sweep512_distinct"] = {
            "workload": "BASELINE configs[4], distinct datasets: 512 fits (alpha x w_smooth grid), each binning its own %d-visibility "
                        "table (%d resident tables in turn, range cache off), N=%d, through the pipeline" % (NV, NT, N),
            "fits_per_s": len(grid) / dt5, "s_total": dt5,
            "iterations_min_median_max": [int(np.min(its5)), int(np.median(its5)), int(np.max(its5))],
            "note": "a pipeline hands out its launches in submission order: a launch of 64 fit loops ends with its longest fit "
                    "(the grid's fits differ 20 x in length), where the shared-(M, j) sweep pauses its stragglers and moves them to clusters"}
        for t in tabs:
            L.lib.fh_vis_destroy(t)
        del f5
    except Exception as e:  # noqa: BLE001
        ex["sweep512_distinct"] = {"error": repr(e)}
    # -- LogNormal fits in one launch, one compute unit each (fh_fit_lognormal_batched): 64 points of an (alpha, w_smooth) grid
    #    over the same 1e6-visibility mapping, default line search
    try:
        al, ws = np.meshgrid(np.linspace(1.2, 1.5, 8), np.logspace(-3, -1, 8))
        al, ws = np.ascontiguousarray(al.ravel()), np.ascontiguousarray(ws.ravel())
        B = al.size
        p0 = np.full(B, 1e-35)
        s_map, pp = np.empty((B, N)), np.empty((B, N))
        niter = (ctypes.c_int * B)()
        status = (ctypes.c_int * B)()
        stats = (ctypes.c_int64 * (9 * B))()
        nv = min(f.n, 1_000_000)
        L.check(L.lib.fh_ctx_set_lognormal_linesearch(f.ctx, 0))
        t0 = time.perf_counter()
        f.bin(nv)
        finalize()
        L.check(L.lib.fh_fit_lognormal_batched(f.ctx, None, None, B, L.ptr(al), L.ptr(p0), L.ptr(ws), h["tol"], h["max_iter"],
                                               1e5, L.ptr(s_map), L.ptr(pp), niter, status, stats))
        dt = time.perf_counter() - t0
        its = np.array(list(niter))
        ex["lognormal_batched64"] = {"workload": "%d LogNormal fits (alpha x w_smooth grid) of one mapping of %d visibilities, "
                                                 "N=%d, staged: every fit on a cluster of four workgroups (64 fits leave three quarters of "
                                                 "the device idle otherwise) until a sixth of them is left, those paused and continued on "
                                                 "clusters of eight" % (B, nv, N),
                                     "linesearch": "linear", "fits_per_s": B / dt, "s_total": dt,
                                     "iterations_min_median_max": [int(its.min()), int(np.median(its)), int(its.max())],
                                     "at_max_iter": int(np.sum(its >= h["max_iter"])),
                                     "failed": int(np.sum(np.array(list(status)) != 0)),
                                     "note": "the points that run to max_iter (alpha >= 1.4: the LogNormal iteration does not converge "
                                             "there) set the time: 2 001 passes heavy in Newton work (4 ms each on one compute unit at "
                                             "the end, 2.5 ms on a cluster of eight); the single launch of rounds 3-5 took 4.4 s for this grid, "
                                             "staged with one compute unit per fit in the first stage 3.6 s "
                                             "(tools/ln_batched64.py: same bits either way)"}
    except Exception as e:
        ex["lognormal_batched64"] = {"error": repr(e)}
    # -- UVDataBinner (next-tier row f4): three streaming passes, 72 algorithmic bytes per row
    try:
        from frank_amd.utilities import UVDataBinner
        n = 10_000_000
        rng = np.random.default_rng(0)
        q = np.exp(rng.uniform(np.log(1e4), np.log(2e6), n))
        V = rng.normal(size=n) + 1j * rng.normal(size=n)
        w = rng.uniform(0.5, 2.0, n)
        best = None
        for _ in range(3):
            b = UVDataBinner(q, V, w, 2e4)
            k = float(L.lib.fh_uvbin_kernel_ms(b._handle))
            best = k if best is None else min(best, k)
        gbps = 72.0 * n / best / 1e6
        ex["uvbin"] = {"workload": "UVDataBinner, %d rows, %d bins" % (n, len(b)), "kernels_ms": best,
                       "roofline": {"bound": "hbm", "achieved": gbps, "peak": 8000.0, "unit": "GB/s",
                                    "frac": gbps / 8000.0, "algorithmic_bytes_per_row": 72}}
    except Exception as e:
        ex["uvbin"] = {"error": repr(e)}
    # -- the geometry fit that calls the hot path inside its residual function (geometry.py:600-763), on the resident
    #    headline table: N = 20, started 5 degrees off; and its residual pass alone against the HBM roof
    try:
        from frank_amd import DiscreteHankelTransform
        from frank_amd.constants import rad_to_arcsec
        from frank_amd.geometry import FitGeometryFourierBessel
        from frank_amd.mock import MOCK_GEOMETRY

        class Resident(object):  # (the table the bench already holds, as frank_amd.geometry._ResidentTable presents one)
            handle, n = f.vis, f.n
        dht = DiscreteHankelTransform(RMAX_ARCSEC / rad_to_arcsec, 20, device=f.device)
        fg = FitGeometryFourierBessel(RMAX_ARCSEC, 20, guess=[30.0, 80.0, 0.0, 0.0])
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            x, ok = fg._fit_on_device(dht, Resident)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        # the reference's own driver on the same residual function (vectors copied out, MINPACK's QR on the host)
        from scipy.optimize import least_squares
        t0 = time.perf_counter()
        sp = least_squares(fg._residual, [30.0, 80.0, 0.0, 0.0], kwargs={"uvdata": (dht, Resident)}, method="lm")
        t_scipy = time.perf_counter() - t0
        g20, I20 = fg._profile_under(fg._trial_geometry(x), dht, Resident)
        ss = ctypes.c_double()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            L.check(L.lib.fh_vis_residuals(dht.context(), ctypes.byref(g20), 0, f.vis, 0, f.n, L.ptr(I20), None, ctypes.byref(ss)))
            ts.append(time.perf_counter() - t0)
        gbps = 40.0 * f.n / min(ts) / 1e9
        ex["geometry_fit"] = {"workload": "FitGeometryFourierBessel(Rmax=%g, N=20) of the %d resident visibilities, started at "
                                          "(30, 80, 0, 0); Levenberg-Marquardt on device-reduced normal equations" % (RMAX_ARCSEC, f.n),
                              "s_per_fit": best, "converged": bool(ok), "inc_PA_dRA_dDec": [float(t) for t in x],
                              "scipy_driver": {"s_per_fit": t_scipy, "inc_PA_dRA_dDec": [float(t) for t in sp.x],
                                               "residual_evaluations": int(sp.nfev),
                                               "max_abs_diff": float(np.abs(np.asarray(x) - sp.x).max())},
                              "note": "the table was made with " + repr([MOCK_GEOMETRY[k] for k in ("inc", "PA", "dRA", "dDec")]) +
                                      "; 20 collocation points reach q = %.2e against baselines to 2e6, hence the bias -- the "
                                      "reference's, both drivers land on it" % dht.q[-1],
                              "residual_pass_ms_wall": 1e3 * min(ts),
                              "roofline_residual_pass": {"bound": "hbm", "achieved": gbps, "peak": 8000.0, "unit": "GB/s",
                                                         "frac": gbps / 8000.0, "algorithmic_bytes_per_row": 40,
                                                         "note": "wall time of the C call (launch, two kernels, one 8-byte copy back)"}}
    except Exception as e:
        ex["geometry_fit"] = {"error": repr(e)}
    return ex


def sharded_leg(f, L, a, dist, rank, world, local_rank, barrier, out, out2):
    """BASELINE configs[3]: ONE fit whose visibilities are sharded over the ranks (contiguous slabs);
    every rank bins its slab, ONE RCCL all-reduce sums the packed upper-triangle Gram + scalars (and a 2-double
    max-reduce the baseline range), then every rank holds M, j and solves (rank 0's solve is the fit)."""
    import torch
    from frank_amd.distributed import make_comm

    def bcast(ident):
        t = torch.tensor(list(ident if ident is not None else bytes(128)), dtype=torch.uint8)
        dist.broadcast(t, 0)
        return bytes(t.tolist())
    # RCCL over xGMI, one rank per GPU; ranks that share a device (more ranks than GPUs: RCCL refuses that) reduce through the
    # host over gloo instead (frank_amd.distributed.HostComm) -- the leg then says so under "comm"
    if world > max(L.device_count(), 1):
        os.environ["FRANK_AMD_COMM"] = "host"
    comm = make_comm(rank, world, f.device, bcast)
    shard = f.n_shard
    times, ar_ms, bin_ms = [], [], []
    nit_s = 0
    for i in range(4):  # the first pass warms RCCL's channels up and is not reported
        barrier()
        t0 = time.perf_counter()
        f.bin(shard)
        comm.allreduce_stats(f.ctx)
        nit_s = f.solve()
        f.sync()
        barrier()
        if i:
            times.append(time.perf_counter() - t0)
            ar_ms.append(comm.last_allreduce_ms())
            bin_ms.append(f.prepass_ms() + f.kernel_ms())
    t = torch.tensor([min(times)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    pass_ms = torch.tensor([float(np.median(bin_ms))], dtype=torch.float64)
    pass_all = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(pass_all, pass_ms)
    payload = ctypes.c_int64(0)
    L.check(L.lib.fh_stats_device(f.ctx, None, ctypes.byref(payload), None))
    nbytes = payload.value * 8
    out.update({"workload": "BASELINE configs[3]: one N=%d fit of %d visibilities sharded over %d ranks "
                            "(%d per rank)" % (a.ncoll, shard * world, world, shard),
                "nvis_total": shard * world, "nvis_per_rank": shard, "rccl_ranks": comm.size(), "comm": type(comm).__name__,
                "s_per_fit": float(t.item()), "fits_per_s": 1.0 / float(t.item()),
                "vis_per_s": shard * world / float(t.item()),
                "allreduce_us": 1e3 * float(np.median(ar_ms)),
                # the payload over the seven xGMI links of a GPU in one shot, plus the latency of a hop
                "allreduce_bound_us": {"payload_bytes": nbytes, "one_shot_over_7_links_us": nbytes / (7 * 153e9) * 1e6,
                                       "note": "%.0f KB is latency-bound: a ring all-reduce makes 2 (ranks - 1) hops of "
                                               "a few microseconds each" % (nbytes / 1024.0)},
                "binning_pass_ms_per_rank": [float(x.item()) for x in pass_all],
                "iterations": nit_s,
                "collective": ("ncclAllReduce(sum) of %d doubles (%.0f KB) + ncclAllReduce(max) of 2 doubles, on the "
                               "context's stream" if type(comm).__name__ == "RcclComm" else
                               "gloo all_reduce(sum) of %d doubles (%.0f KB) + all_reduce(max) of 2 doubles, staged through the "
                               "host (the ranks share a device)") % (payload.value, nbytes / 1024.0)})
    # -- BASELINE configs[4] over the ranks: 512 fits (32 alpha x 16 w_smooth) of ONE mapping of 1e6 visibilities.  Rank 0
    #    bins the table; the packed statistics (380 KB) reach every rank through the same all-reduce (the other ranks
    #    contribute zeros); every rank runs its contiguous slice of the grid in one batched launch, no further communication
    #    (the reference's loop: frank/fit.py:534-548)
    try:
        h = HYPER
        al, ws = np.meshgrid(np.linspace(1.01, 1.5, 32), np.logspace(-4, -1, 16))
        al, ws = np.ascontiguousarray(al.ravel()), np.ascontiguousarray(ws.ravel())
        B = al.size
        from frank_amd.distributed import shard_range
        first, count = shard_range(B, rank, world)
        nv = min(f.n, 1_000_000)
        mu, pp = np.empty((max(count, 1), a.ncoll)), np.empty((max(count, 1), a.ncoll))
        niter = (ctypes.c_int * max(count, 1))()
        status = (ctypes.c_int * max(count, 1))()
        best, per_rank = None, None
        for i in range(2):
            barrier()
            t0 = time.perf_counter()
            L.check(L.lib.fh_bin_reset(f.ctx))
            if rank == 0:
                L.check(L.lib.fh_bin_visibilities(f.ctx, ctypes.byref(f.geom), f.vis, 0, nv))
            comm.allreduce_stats(f.ctx)
            L.check(L.lib.fh_stats_finalize(f.ctx, ctypes.byref(f.geom), 0, 0, None, None, None, None, None))
            if count:
                p0 = np.full(count, h["p0"])
                L.check(L.lib.fh_fit_normal_batched(f.ctx, None, None, count, L.ptr(np.ascontiguousarray(al[first:first + count])),
                                                    L.ptr(p0), L.ptr(np.ascontiguousarray(ws[first:first + count])), h["tol"],
                                                    h["max_iter"], L.ptr(mu), L.ptr(pp), niter, status))
            f.sync()
            mine = time.perf_counter() - t0
            barrier()
            dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            if best is None or float(dt.item()) < best:
                best = float(dt.item())
                mt = torch.tensor([mine], dtype=torch.float64)
                allm = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
                dist.all_gather(allm, mt)
                per_rank = [float(x.item()) for x in allm]
        bad = torch.tensor([int(np.sum(np.array(list(status))[:count] != 0))], dtype=torch.int64)
        dist.all_reduce(bad, op=dist.ReduceOp.SUM)
        out2.update({"workload": "BASELINE configs[4]: %d fits (alpha x w_smooth grid), N=%d, one mapping of %d visibilities "
                                 "binned on rank 0, statistics to all ranks by one RCCL all-reduce, %d fits per rank"
                                 % (B, a.ncoll, nv, -(-B // world)),
                     "fits_per_s": B / best, "s_total": best, "per_rank_s": per_rank, "rccl_ranks": comm.size(),
                     "failed": int(bad.item())})
    except BaseException as e:  # noqa: BLE001
        out2["error"] = repr(e)
    comm.close()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    # The launches of the fit loops take turns on four HIP streams beside the binning stream; HIP multiplexes streams onto
    # GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels sharing a queue serialise: more queues than streams, so that the
    # streams of the several contexts of this script never share one (the library sets the same default when it is loaded).
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
    # load the HIP library BEFORE torch so that ROCm's own runtime libraries serve the process
    from frank_amd import _lib as L
    from frank_amd.mock import mock_disc_visibilities

    dist = None
    if world > 1 or a.force_legs:
        import torch
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        # (gloo announces its mesh on STDOUT from C++ -- "[Gloo] Rank 3 is connected to 7 peer ranks ..." -- from every rank and no
        #  environment switch turns that off; this program's stdout is ONE JSON line: file descriptor 1 is pointed at stderr while
        #  the group forms and while the first barrier runs.  That barrier comes AFTER this process has created its device context:
        #  torch.distributed probes torch's own HIP runtime in it, and a process whose first HIP call went to that runtime finds
        #  no device through the library's)
        with _stdout_to_stderr():
            dist.init_process_group("gloo", rank=rank, world_size=world)

    first_barrier = [dist is not None]

    def barrier():
        if dist is not None:
            if first_barrier[0]:
                first_barrier[0] = False
                with _stdout_to_stderr():
                    dist.barrier()
            else:
                dist.barrier()

    ndev = max(L.device_count(), 1)
    f = Fitter(L, a.ncoll, local_rank % ndev)  # (more ranks than GPUs only happens in the 1-GPU smoke run of this path)
    do_shard = (world > 1 or a.force_legs) and not a.no_sharded
    f.nfit = a.nvis
    f.n_shard = int(min(-(-int(a.sharded_total) // world), int(a.sharded_cap))) if do_shard else 0
    nrows = max(a.nvis, f.n_shard)
    u, v, V, w = mock_disc_visibilities(nrows, seed=1000 * rank, noise_seed=50 + rank)
    f.upload(u, v, V, w)
    # the ring of the headline: three more resident tables -- separate table objects holding the SAME rows, so that every step of the
    # timed region is the fit BASELINE configs[1] names (the reference's input: 667 iterations; tables drawn from other seeds take
    # 634 .. 776 and the region ends with its longest fit) -- and every step bins the next one with the baseline-range cache off:
    # a table object the context has not binned last, nothing it learned from an earlier step is used
    RING = 4
    for k in range(1, RING):
        keep, keep_n = f.vis, f.n
        f.upload(u[:a.nvis], v[:a.nvis], V[:a.nvis], w[:a.nvis])
        f.vis, f.n = keep, keep_n
    del u, v, V, w
    ring = f.tables[:RING]
    L.check(L.lib.fh_ctx_set_range_cache(f.ctx, 0))

    if a.prewarm_seconds > 0:  # (untimed; see --prewarm-seconds)
        t_pre = time.perf_counter()
        while time.perf_counter() - t_pre < a.prewarm_seconds:
            f.run_steps(20, ring=ring)
            f.sync()
    f.run_steps(a.warmup, ring=ring)
    f.sync()
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    f.run_steps(a.steps, kernel_ms, ring=ring)
    f.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    L.check(L.lib.fh_ctx_set_range_cache(f.ctx, 1))
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    fallbacks_timed = f.cluster_info()[1]
    # split of one step (untimed, after the measured region): the fit loop as a single fit runs it (a cluster of workgroups,
    # include/frank_hip.h: fh_fit_cluster_info) and on ONE compute unit, the form the steady state runs
    # (the pass as the timed region runs it: nothing remembered, the range kernel looked ahead)
    L.check(L.lib.fh_ctx_set_range_cache(f.ctx, 0))
    L.check(L.lib.fh_bin_prefetch_range(f.ctx, ctypes.byref(f.geom), ring[1], 0, f.nfit))
    f.bin(vis=ring[1]); f.sync()
    L.check(L.lib.fh_bin_prefetch_range(f.ctx, ctypes.byref(f.geom), f.vis, 0, f.nfit))
    f.sync()
    t0 = time.perf_counter(); f.bin(); f.sync(); t_bin = time.perf_counter() - t0
    L.check(L.lib.fh_ctx_set_range_cache(f.ctx, 1))
    t0 = time.perf_counter(); nit = f.solve(); f.sync(); t_solve = time.perf_counter() - t0  # (table 0: the reference's input)
    kms_alone = f.kernel_ms()
    rng_alone = f.range_ms()
    pre_alone = f.prepass_ms() + rng_alone
    loop_ms = f.loop_kernel_ms()
    loop_wgs = f.cluster_info()[0]
    prev = os.environ.get("FRANK_AMD_K2_CLUSTER")
    os.environ["FRANK_AMD_K2_CLUSTER"] = "1"
    f.solve(); f.sync()
    loop_ms_one = f.loop_kernel_ms()
    if prev is None:
        del os.environ["FRANK_AMD_K2_CLUSTER"]
    else:
        os.environ["FRANK_AMD_K2_CLUSTER"] = prev

    sharded = sweep_multi = None
    hung = False
    if do_shard:
        # in a watchdog thread: an RCCL failure or hang must cost this key, never the headline line
        import threading
        sharded, sweep_multi = {}, {}

        def leg():
            try:
                sharded_leg(f, L, a, dist, rank, world, local_rank, barrier, sharded, sweep_multi)
            except BaseException as e:  # noqa: BLE001
                sharded["error"] = repr(e)
                sweep_multi.setdefault("error", "the sharded leg before it failed: " + repr(e))
        th = threading.Thread(target=leg, daemon=True)
        th.start()
        th.join(timeout=240.0)
        if th.is_alive():
            hung = True
            sharded = {"error": "sharded leg did not finish within 240 s (rank %d)" % rank}
            sweep_multi = {"error": "not reached"}

    if rank == 0:
        fits = a.steps * world
        value = fits / elapsed
        kms = float(np.mean(kernel_ms))
        Nc = a.ncoll
        # -- the dominant kernel: fit_loop_kernel, ONE workgroup = one CU per fit.  Algorithmic flops of a pass (DESIGN.md
        #    K2): Cholesky of the (N+1) x (N+1) augmented precision, n^3 / 3, + inverse of the triangular factor, n^3 / 3;
        #    the kernel makes iterations + 2 passes (the two seed solves of radial_fitters.py:744-752)
        n_aug = Nc + 1
        flops_pass = 2.0 * n_aug ** 3 / 3.0
        flops_fit = flops_pass * (nit + 2)
        achieved = flops_fit / (loop_ms * 1e-3) / 1e12
        achieved_one = flops_fit / (loop_ms_one * 1e-3) / 1e12
        peak_cu = FP64_MFMA_PEAK_TFLOPS / N_CU
        # HBM bytes per launch: STATIC values, read from the PMC summaries committed under profiles/ (rocprofv3 cannot run inside
        # bench.py); they describe the build the profile was taken from, named in the *_source fields
        traffic, traffic_one, traffic_src, bin_traffic, bin_traffic_src, profile_lib = None, None, None, None, None, None

        def pmc(name):  # (the newest profile that has been taken: round 6, 5, 4)
            for nm in (name.replace("r04_", "r06_"), name.replace("r04_", "r05_"), name):
                try:
                    with open(os.path.join(ROOT, "profiles", nm)) as fh:
                        d = json.load(fh)
                    d["_file"] = nm
                    return d
                except OSError:
                    continue
            raise OSError(name)
        try:
            pm2, pm1 = pmc("r04_pmc_fit_loop_cluster.json"), pmc("r04_pmc_fit_loop.json")
            profile_lib = pm2.get("_library")
            if Nc == 300:
                # (the profiled fit is the 1e6-visibility fixture: 825 passes; scaled to the passes of this fit)
                traffic = int([e for k, e in pm2.items() if k.startswith("fit_loop_kernel")][0]["hbm_bytes_per_launch"] * (nit + 2) / 825.0)
                traffic_one = int([e for k, e in pm1.items() if k.startswith("fit_loop_kernel")][0]["hbm_bytes_per_launch"] * (nit + 2) / 825.0)
            traffic_src = ("static: profiles/%s / %s (rocprofv3 --pmc FETCH_SIZE / "
                           "WRITE_SIZE on one fit of 825 passes, scaled to this fit's passes; FETCH doubled per the gfx950 note), "
                           "taken from the library build '%s'; not measured in this run" % (pm2.get("_file"), pm1.get("_file"), profile_lib))
        except Exception:
            traffic = None
        try:
            pm, pmf = pmc("r04_pmc_binning.json"), pmc("r04_pmc_binning_first_sight.json")
            if Nc == 300 and a.nvis == N_VIS:
                # a pass over rows the context binned last keeps the (u, v) histogram and its scan: those two kernels do not run
                # the pass of the timed region looks at (u, v) twice (range, histogram) and scans: every kernel of the first-sight profile
                bin_traffic = {k: int(e["hbm_bytes_per_launch"]) for k, e in pmf.items() if isinstance(e, dict) and "hbm_bytes_per_launch" in e}
                cached = sum(int(e["hbm_bytes_per_launch"]) for k, e in pm.items()
                             if isinstance(e, dict) and "hbm_bytes_per_launch" in e and not k.startswith(("uv_hist", "bucket_scan")))
                bin_traffic["total"] = int(sum(bin_traffic.values()))
                bin_traffic["total_with_the_caches_on"] = int(cached)
            bin_traffic_src = ("static: profiles/%s, %s (one pass of 1e7 visibilities "
                               "at N = 300, per kernel; 2 x FETCH_SIZE + WRITE_SIZE), library build '%s'; not measured in this run"
                               % (pm.get("_file"), pmf.get("_file"), pm.get("_library")))
        except Exception:
            bin_traffic = None
        # the reference's own run of this very input (tests/golden/fit_N300_1e7.npz: 667 iterations) -- rank 0, default sizes
        ref_iters = None
        try:
            if Nc == 300 and a.nvis == N_VIS:
                ref_iters = int(np.load(os.path.join(ROOT, "tests", "golden", "fit_N300_1e7.npz"))["niter"])
        except Exception:
            ref_iters = None
        # -- the binning pass (deproject .. sort .. moments .. Gram of the compressed rows), HBM-bound: 40 B per visibility
        #    (u, v, Re V, Im V, w) over the time of the whole pass, by events
        pass_ms = pre_alone + kms_alone
        bin_GBps = 40.0 * a.nvis / (pass_ms * 1e-3) / 1e9
        flops_sym = a.nvis * (Nc * (Nc + 1) + 2 * Nc)       # what binning the visibilities row by row costs
        out = {
            "metric": "FrankFitter solves/sec (N=%d, %.0e visibilities per fit, Normal, fp64, end-to-end)" % (Nc, a.nvis),
            "library": L.lib.fh_version().decode(),
            "value": value, "unit": "fits/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: N=%d, %d mock-disc visibilities resident in HBM, Normal GP "
                                   "fit, one independent fit per GPU per step; the steps take turns on a ring of %d "
                                   "resident table objects (the same rows: every step is the reference's input) with the "
                                   "baseline-range cache off: nothing is remembered between steps" % (Nc, a.nvis, RING),
                       "alpha": HYPER["alpha"], "wsmooth": HYPER["wsmooth"], "tol": HYPER["tol"],
                       "iterations_to_converge": nit, "iterations_of_the_reference_on_this_input": ref_iters,
                       "iterations_match_the_reference": (nit == ref_iters) if ref_iters is not None else None,
                       "parallelism": "independent fits x%d" % world,
                       "prewarm_seconds": a.prewarm_seconds,
                       "prewarm": "untimed run of the same pipeline in front of the W warm-up steps (a device that has been idle "
                                  "reaches its clocks over its first seconds of load); --prewarm-seconds 0 turns it off"},
            "breakdown_ms": {"single_fit_latency": 1e3 * (t_bin + t_solve), "bin_gram_pass": 1e3 * t_bin,
                             "finalize_plus_iterate": 1e3 * t_solve, "us_per_iteration": 1e6 * t_solve / max(nit, 1),
                             "binning_pass_by_events": pre_alone + kms_alone, "fit_loop_kernel": loop_ms,
                             "fit_loop_workgroups": loop_wgs, "fit_loop_kernel_on_one_cu": loop_ms_one,
                             "us_per_pass_cluster": 1e3 * loop_ms / (nit + 2), "us_per_pass_one_cu": 1e3 * loop_ms_one / (nit + 2),
                             "cluster_fallbacks_in_the_timed_region": fallbacks_timed,
                             "note": "steps are pipelined: fit i's iteration overlaps fit i+1's binning; a shallow pipeline (this "
                                     "region) runs every fit on a cluster of workgroups of one XCD (fit_loop.hip, clu::), a deep "
                                     "one (extra.steady_state) on one compute unit each; the timed region = steps x (binning + "
                                     "hand-over) + one drain of finalize_plus_iterate"},
            "roofline": {"kernel": K2_KERNEL_NAME + " (cluster mode: %d workgroups per fit, as the timed region runs it)" % loop_wgs,
                         "bound": "mfma", "achieved": achieved, "peak": peak_cu * loop_wgs, "unit": "TFLOP/s",
                         "frac": achieved / (peak_cu * loop_wgs), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_ms": loop_ms, "passes": nit + 2, "algorithmic_flops_per_pass": flops_pass,
                         "peak_note": "peak = the share of the 78.6 TFLOP/s fp64 matrix peak of the %d compute units a fit holds "
                                      "in cluster mode (%d/256); the mode buys latency (a pass is a chain of 19 dependent tile "
                                      "factorisations) with compute units that mostly wait -- the fraction per CU is lower "
                                      "than on one CU by design" % (loop_wgs, loop_wgs),
                         "why_this_kernel": "most of the GPU time of the timed region (profiles/r06_kernel_stats.csv)",
                         "traffic_one_cu": traffic_one, "profile_library": profile_lib,
                         "one_cu": {"kernel": K2_KERNEL_NAME + " on one compute unit (FRANK_AMD_K2_CLUSTER=1: the form of the steady "
                                              "state and of the batched sweeps)", "achieved": achieved_one, "peak": peak_cu,
                                    "frac": achieved_one / peak_cu, "kernel_ms": loop_ms_one}},
            "roofline_binning": {"kernel": "binning pass of rows the context has not binned last: uv_hist (range + histograms: one look), bucket_scan, deproject_scatter, piece_moments, bucket_factor2, "
                                           + K1_KERNEL_NAME + ", vr_finish (bin_prepass.hip)",
                                 "bound": "hbm", "achieved": bin_GBps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": bin_GBps / HBM_PEAK_GBPS, "pass_ms": pass_ms, "gram_kernel_ms": kms_alone, "range_kernel_ms": rng_alone,
                                 "algorithmic_bytes_per_vis": 40, "traffic": bin_traffic, "traffic_source": bin_traffic_src,
                                 "achieved_on_traffic_GBps": (bin_traffic["total"] / (pass_ms * 1e-3) / 1e9) if bin_traffic else None,
                                 "row_by_row_equivalent_TFLOPs": flops_sym / (pass_ms * 1e-3) / 1e12,
                                 "note": "the rows of a J0 bucket enter the Gram through 12 x 12 moments, so the pass is "
                                         "memory-bound; row_by_row_equivalent is what binning every visibility on the "
                                         "matrix pipe (the rows kernel, 15.2 ms = 0.76 of the fp64 matrix peak) would need; "
                                         "what the pass moves: 16 B (u, v: ONE look gives the range and the histograms -- on the look-ahead stream) + "
                                         "40 B (all columns) + 24 B written + 24 B read = 104 B per visibility when nothing is remembered, as the "
                                         "timed region runs it (88 B with the histograms kept: extra.headline_with_caches); the fused "
                                         "one-pass form, 42 B per visibility, is slower: profiles/r06_binning_fused.txt; pass_ms = the range "
                                         "kernel (on the look-ahead stream) + the pre-pass + the Gram kernels, by events"},
        }
        if ref_iters is not None and nit != ref_iters:
            out["parity_error"] = "the fit took %d iterations, the reference %d on the same input" % (nit, ref_iters)
        if sharded is not None:
            out["sharded_fit"] = sharded
            out["sweep512_multi"] = sweep_multi
            out["multi_gpu_note"] = ("the builder's box has one GPU: no N > 1 value of this line was ever measured by the "
                                     "builder; the scaling curve is the driver's")
        if world == 1 and not a.no_extras:
            out["extra"] = extras(f, L, a)
            ss = out["extra"].get("steady_state", {})
            if "fits_per_s" in ss:  # the same algorithmic flops against the WHOLE chip, at the rate the pipeline sustains
                out["roofline"]["chip_fraction_at_steady_state"] = ss["fits_per_s"] * flops_fit / 1e12 / FP64_MFMA_PEAK_TFLOPS
                out["roofline"]["chip_note"] = ("at steady state a fit loop is one CU and up to %d fits are outstanding, in "
                                                "launches of up to 64 fit loops (~150 loops resident: fits/s x time per fit), which from 128 "
                                                "fits in flight on keep their matrix in registers (fit_loop_rr.hip: 0.23 MB per pass beyond the "
                                                "L2 instead of 5.0); the rate is set by what a pass of that form costs (instruction issue, 150-160 "
                                                "us) and by the compute units the binning passes take" % ss.get("fit_slots", 0))
        if not a.no_cpu_baseline and world == 1:  # the CPU leg is timed at N=1 only
            out["cpu_baseline"] = cpu_baseline(a.ncoll, a.nvis, nit)
        print(json.dumps(out), flush=True)
    if hung:
        # a collective that never returned cannot be torn down cleanly; the line (with the leg's error under its key) is out,
        # and a process that has touched the GPU and gives up says so in its exit code
        sys.stdout.flush()
        os._exit(3)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

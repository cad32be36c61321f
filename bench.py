#!/usr/bin/env python3
"""bench.py -- FrankFitter solves/sec on MI355X (BASELINE.json metric).

A "step" is ONE complete fit, end to end on the device, of the configuration the metric is quoted on
(BASELINE.json configs[1]): N = 300 collocation points, 1e7 synthetic mock-disc visibilities already
resident in HBM, Normal GP fit, fp64:
    bin_gram (deproject + J0 design block + Gram)  ->  [RCCL all-reduce in --mode shard]
    ->  scale/unpack M, j  ->  the full power-spectrum iteration to convergence (tol 1e-3).
Nothing is cached between steps.  `value` = fits completed by all ranks / max-over-ranks wall time.

    python bench.py                                  # 1 GPU, defaults finish in a few minutes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: fits are independent objects, so each rank fits its own 1e7-visibility dataset (weak
scaling, no data-path collective).  The sharded-visibility path with the RCCL all-reduce of the
(N^2+N)-sized sufficient statistics (BASELINE.json configs[3]) is timed separately and reported under
"sharded_fit".  torch is used ONLY for the rendezvous / barrier / max-over-ranks (gloo, CPU tensors);
the data path is libfrank_hip + RCCL.
"""
import argparse
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # the timed region holds K binning passes back to back plus ONE pipeline drain (the iteration of the last fit,
    # ~0.24 s): the default K amortises it, a small K mostly measures it
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nvis", type=int, default=N_VIS)
    ap.add_argument("--ncoll", type=int, default=N_COLL)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sharded", action="store_true",
                    help="also time BASELINE configs[3] (one fit sharded over the ranks, RCCL all-reduce); opt-in so "
                         "that a collective problem can never cost the headline line")
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
        self.n = 0

    def upload(self, u, v, V, w):
        L = self.L
        vis = ctypes.c_void_p()
        Vre, Vim = np.ascontiguousarray(V.real), np.ascontiguousarray(V.imag)
        L.check(L.lib.fh_vis_upload(self.device, L.ptr(u), L.ptr(v), L.ptr(Vre), L.ptr(Vim), L.ptr(w), w.size, u.size,
                                    ctypes.byref(vis)))
        self.vis, self.n = vis, u.size

    def bin(self):
        L = self.L
        L.check(L.lib.fh_bin_reset(self.ctx))
        L.check(L.lib.fh_bin_visibilities(self.ctx, ctypes.byref(self.geom), self.vis, 0, self.n))

    def kernel_ms(self):
        ms = ctypes.c_float(0)
        self.L.check(self.L.lib.fh_bin_last_kernel_ms(self.ctx, ctypes.byref(ms)))
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

    def submit(self):
        """bin_gram on the main stream, then hand the iteration to a fit slot (fit_loop kernel on its own stream)."""
        L = self.L
        self.bin()
        H0, qmn, qmx = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        L.check(L.lib.fh_stats_finalize(self.ctx, ctypes.byref(self.geom), 0, 1, None, None, ctypes.byref(H0),
                                        ctypes.byref(qmn), ctypes.byref(qmx)))
        h = HYPER
        t = ctypes.c_int(-1)
        L.check(L.lib.fh_fit_submit(self.ctx, h["alpha"], h["p0"], h["wsmooth"], h["tol"], h["max_iter"],
                                    ctypes.byref(t)))
        return t.value

    def collect(self, ticket):
        L = self.L
        L.check(L.lib.fh_fit_collect(self.ctx, ticket, L.ptr(self.mu), L.ptr(self.p), ctypes.byref(self.niter)))
        return self.niter.value

    def run_steps(self, k, kernel_ms=None):
        """k independent end-to-end fits, pipelined: the iteration of fit i overlaps the binning of fit i+1."""
        L = self.L
        slots = L.lib.fh_fit_slots()
        pending, nit = [], 0
        for _ in range(k):
            if len(pending) == slots:
                nit = self.collect(pending.pop(0))
            pending.append(self.submit())
            if kernel_ms is not None:
                kernel_ms.append(self.kernel_ms())
        L.check(L.lib.fh_fit_flush(self.ctx))  # the last, partly filled launch
        for t in pending:
            nit = self.collect(t)
        return nit

    def sync(self):
        self.L.check(self.L.lib.fh_ctx_synchronize(self.ctx))


def cpu_baseline(ncoll, nvis, gpu_niter):
    """The CPU oracle (oracle/frank_oracle.c, a single-threaded port of the reference path) on a bounded sample."""
    from frank_amd.constants import rad_to_arcsec
    from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities
    from oracle import oracle as fo
    g = MOCK_GEOMETRY
    geom = (g["inc"], g["PA"], g["dRA"], g["dDec"])
    ns = min(nvis, 100_000 if ncoll >= 200 else 400_000)
    u, v, V, w = mock_disc_visibilities(ns, seed=0, noise_seed=50)
    t0 = time.perf_counter()
    m = fo.map_visibilities(ncoll, RMAX_ARCSEC / rad_to_arcsec, geom, u, v, V, w)
    t_map = time.perf_counter() - t0
    it = 150
    t0 = time.perf_counter()
    out = fo.frank_fit_normal(ncoll, RMAX_ARCSEC / rad_to_arcsec, m["M"], m["j"], max_iter=it, **{
        k: HYPER[k] for k in ("alpha", "p0", "wsmooth", "tol")})
    t_it = (time.perf_counter() - t0) / max(out["niter"], 1)
    t_fit_total = t_map * (nvis / ns) + t_it * gpu_niter
    return {"value": 1.0 / t_fit_total, "unit": "fits/s", "cores": 1, "kind": "port",
            "sample": "oracle map_visibilities on %d of %d visibilities (%.1f s, scaled linearly) + %d of the %d "
                      "power-spectrum iterations (%.1f ms/iteration, scaled)" % (ns, nvis, t_map, out["niter"],
                                                                               gpu_niter, 1e3 * t_it),
            "s_per_fit": t_fit_total}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    # Every outstanding fit_loop kernel lives on its own HIP stream; HIP multiplexes streams onto
    # GPU_MAX_HW_QUEUES hardware queues (default 4) and kernels sharing a queue serialise.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
    # load the HIP library BEFORE torch so that ROCm's own runtime libraries serve the process
    from frank_amd import _lib as L
    from frank_amd.mock import mock_disc_visibilities

    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if dist is not None:
            dist.barrier()

    ndev = max(L.device_count(), 1)
    f = Fitter(L, a.ncoll, local_rank % ndev)  # (more ranks than GPUs only happens in the 1-GPU smoke run of this path)
    u, v, V, w = mock_disc_visibilities(a.nvis, seed=1000 * rank, noise_seed=50 + rank)
    f.upload(u, v, V, w)
    del u, v, V, w

    f.run_steps(a.warmup)
    f.sync()
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    nit = f.run_steps(a.steps, kernel_ms)
    f.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # split of one step (untimed, after the measured region)
    t0 = time.perf_counter(); f.bin(); f.sync(); t_bin = time.perf_counter() - t0
    t0 = time.perf_counter(); f.solve(); f.sync(); t_solve = time.perf_counter() - t0
    kms_alone = f.kernel_ms()

    sharded = None
    if world > 1 and a.sharded:
        # BASELINE configs[3]: ONE fit whose visibilities are sharded over the ranks; RCCL all-reduce of the packed
        # upper-triangle Gram + scalars, then every rank holds M, j (rank 0's solve is the fit).
        import torch
        from frank_amd.distributed import RcclComm

        def bcast(ident):
            t = torch.tensor(list(ident if ident is not None else bytes(128)), dtype=torch.uint8)
            dist.broadcast(t, 0)
            return bytes(t.tolist())
        comm = RcclComm(rank, world, local_rank, bcast)
        times = []
        for i in range(3):
            barrier()
            t0 = time.perf_counter()
            f.bin()
            comm.allreduce_stats(f.ctx)
            nit_s = f.solve()
            f.sync()
            barrier()
            times.append(time.perf_counter() - t0)
        sharded = {"nvis_total": a.nvis * world, "s_per_fit": min(times), "iterations": nit_s,
                   "collective": "RCCL all-reduce, %d doubles" % (190 * 256 + 2 if a.ncoll > 207 else 0)}
        comm.close()

    if rank == 0:
        fits = a.steps * world
        value = fits / elapsed
        kms = float(np.mean(kernel_ms))
        Nc = a.ncoll
        flops_sym = a.nvis * (Nc * (Nc + 1) + 2 * Nc)       # SURVEY 8(d) symmetric-half figure (unique outputs)
        flops_full = a.nvis * (2 * Nc * Nc + 2 * Nc)         # SURVEY 8(d) full figure
        achieved = flops_sym / (kms * 1e-3) / 1e12
        traffic = None
        try:  # HBM bytes of one bin_gram launch from the committed PMC passes (rocprofv3 cannot run inside bench.py)
            with open(os.path.join(ROOT, "profiles", "r01_pmc_hbm.json")) as fh:
                traffic = json.load(fh)["bin_gram_kernel"]["hbm_bytes_per_launch"] * (a.nvis / 1e7) if Nc == 300 else None
        except Exception:
            traffic = None
        out = {
            "metric": "FrankFitter solves/sec (N=%d, %.0e visibilities per fit, Normal, fp64, end-to-end)" % (Nc, a.nvis),
            "value": value, "unit": "fits/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: N=%d, %d mock-disc visibilities resident in HBM, Normal GP "
                                   "fit, one independent fit per GPU per step" % (Nc, a.nvis),
                       "alpha": HYPER["alpha"], "wsmooth": HYPER["wsmooth"], "tol": HYPER["tol"],
                       "iterations_to_converge": nit, "parallelism": "independent fits x%d" % world},
            "breakdown_ms": {"single_fit_latency": 1e3 * (t_bin + t_solve), "bin_gram_pass": 1e3 * t_bin,
                             "finalize_plus_iterate": 1e3 * t_solve, "us_per_iteration": 1e6 * t_solve / max(nit, 1),
                             "bin_gram_kernel_alone": kms_alone,
                             "note": "steps are pipelined: fit i's iteration (one CU) overlaps fit i+1's binning; the "
                                     "timed region = steps x (binning + hand-over) + one drain of finalize_plus_iterate"},
            "roofline": {"kernel": "bin_gram_kernel<19>", "bound": "mfma", "achieved": achieved,
                         "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                         "traffic": traffic, "traffic_source": "profiles/r01_pmc_hbm.json (rocprofv3 --pmc FETCH_SIZE / "
                         "WRITE_SIZE, FETCH doubled per the gfx950 note)", "kernel_ms": kms, "kernel_ms_min_median_max": [round(float(x), 2) for x in (np.min(kernel_ms), np.median(kernel_ms), np.max(kernel_ms))],
                         "algorithmic_flops_per_vis": Nc * (Nc + 1) + 2 * Nc,
                         "achieved_full_gram_equiv": flops_full / (kms * 1e-3) / 1e12,
                         "hbm_read_GBps": 40.0 * a.nvis / (kms * 1e-3) / 1e9 * (1.0 if Nc <= 207 else 2.0)},
        }
        if sharded:
            out["sharded_fit"] = sharded
        if not a.no_cpu_baseline and world == 1:  # the CPU leg is timed at N=1 only
            out["cpu_baseline"] = cpu_baseline(a.ncoll, a.nvis, nit)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""One rank of the two-process sharded fit (tests/test_gpu_configs.py::test_two_processes_sharded_fit_on_this_box), launched by
`python -m torch.distributed.run --nproc-per-node 2 tests/dist_worker.py CASE OUT_DIR`.

BASELINE configs[3] in small: every rank holds a contiguous slab of ONE visibility table (frank_amd.distributed.shard_range), bins
it on its device -- device RANK when the box has as many GPUs as ranks (RCCL over xGMI), else device 0 for everybody with the
reduction staged through the host over gloo (FRANK_AMD_COMM=host: RCCL refuses two ranks on one device) --, the packed statistics
are all-reduced (statistical_models.py:210-211, 218 are the sums being distributed), and every rank finalises M, j and runs the
fit.  Results go to OUT_DIR/rankR.npz; the test compares them with the unsharded fit.
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {  # N, rows, vis_model, scale height
    "tiles_N300": (300, 200001, "opt_thick", None),
    "wide_N320": (320, 60001, "opt_thick", None),
    "debris_N40": (40, 6001, "debris", lambda r: 0.05 + 0.02 * r),
    # the reference's own headline run (tests/golden/fit_N300_1e7.npz: seeds 0 / 50), 5e6 rows per rank
    "ref_N300_1e7": (300, 10000000, "opt_thick", None),
    # the per-GPU share of BASELINE configs[3] (1e8 rows over eight GPUs): two ranks of 1.25e7 rows each
    "share_2x1p25e7": (300, 25000000, "opt_thick", None),
}
SEEDS = {"ref_N300_1e7": (0, 50), "share_2x1p25e7": (0, 50)}
HYPER = dict(alpha=1.05, p0=1e-15, wsmooth=1e-4, tol=1e-3, max_iter=2000)


def main():
    case, out_dir = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    from frank_amd import _lib, FixedGeometry, FourierBesselFitter  # the HIP library before torch
    from frank_amd.distributed import make_comm, shard_range
    from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ndev = max(_lib.device_count(), 1)
    if ndev < world:
        os.environ["FRANK_AMD_COMM"] = "host"
    device = rank % ndev
    N, nvis, vis_model, sh = CASES[case]
    sd, nsd = SEEDS.get(case, (41, 42))
    u, v, V, w = mock_disc_visibilities(nvis, seed=sd, noise_seed=nsd)
    kw = dict(verbose=False, device=device)
    if vis_model == "debris":
        kw.update(assume_optically_thick=False, scale_height=sh)
    geom = FixedGeometry(**MOCK_GEOMETRY)
    FB = FourierBesselFitter(2.0, N, geom, **kw)
    ctx, vm = FB._DHT.context(), FB._vis_map
    _lib.check(_lib.lib.fh_ctx_set_scale_height(ctx, _lib.ptr(_lib.f8(vm._H2)) if vis_model == "debris" else None))
    first, count = shard_range(nvis, rank, world)
    sl = slice(first, first + count)
    uu, vv, ww = (np.ascontiguousarray(x[sl]) for x in (u, v, w))
    Vre, Vim = np.ascontiguousarray(V.real[sl]), np.ascontiguousarray(V.imag[sl])
    vis = ctypes.c_void_p()
    _lib.check(_lib.lib.fh_vis_upload(device, _lib.ptr(uu), _lib.ptr(vv), _lib.ptr(Vre), _lib.ptr(Vim), _lib.ptr(ww), ww.size,
                                      uu.size, ctypes.byref(vis)))

    def bcast(ident):
        t = torch.tensor(list(ident if ident is not None else bytes(128)), dtype=torch.uint8)
        dist.broadcast(t, 0)
        return bytes(t.tolist())
    comm = make_comm(rank, world, device, bcast)
    gm = _lib.make_geometry(geom)
    _lib.check(_lib.lib.fh_bin_reset(ctx))
    _lib.check(_lib.lib.fh_bin_visibilities(ctx, ctypes.byref(gm), vis, 0, count))
    comm.allreduce_stats(ctx)
    M, j = np.empty((N, N)), np.empty(N)
    H0, qmn, qmx = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    _lib.check(_lib.lib.fh_stats_finalize(ctx, ctypes.byref(gm), _lib.VIS_MODELS[vm._vis_model], 1, _lib.ptr(M), _lib.ptr(j),
                                          ctypes.byref(H0), ctypes.byref(qmn), ctypes.byref(qmx)))
    mu, p, nit = np.empty(N), np.empty(N), ctypes.c_int(0)
    h = HYPER
    _lib.check(_lib.lib.fh_fit_normal(ctx, None, None, h["alpha"], h["p0"], h["wsmooth"], h["tol"], h["max_iter"], _lib.ptr(mu),
                                      _lib.ptr(p), ctypes.byref(nit), None, None))
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), M=M, j=j, H0=H0.value, qmin=qmn.value, qmax=qmx.value, mu=mu, p=p,
             niter=nit.value, ranks=comm.size(), kind=type(comm).__name__, device=device, rows=count)
    comm.close()
    _lib.lib.fh_vis_destroy(vis)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Multi-GPU helpers: the visibility axis shards, the (N x N + N) sufficient statistics all-reduce.

M, j and H0 are plain sums over visibilities (statistical_models.py:210-211, 218) and the q-range check
needs min / max (statistical_models.py:512-535), so rank r bins rows shard_range(n, r, world) and one
all-reduce finishes the mapping.  On GPUs the payload is the packed upper-triangle Gram held by the
context (fh_comm_allreduce_stats, RCCL over xGMI); `allreduce_mapping` is the same reduction on host
arrays through any torch.distributed backend (gloo in the CPU tests).
"""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous, near-equal slab [first, first + count) of n rows for `rank` of `world`."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank / world")
    base, rem = divmod(int(n), int(world))
    first = rank * base + min(rank, rem)
    count = base + (1 if rank < rem else 0)
    return first, count


def allreduce_mapping(M, j, H0, qmin, qmax, group=None):
    """Sum M, j, H0 and min/max the baseline range over the ranks of a torch.distributed group (in place safe)."""
    import torch
    import torch.distributed as dist
    N = j.shape[0]
    buf = torch.from_numpy(np.concatenate([np.asarray(M, dtype=np.float64).reshape(-1),
                                           np.asarray(j, dtype=np.float64), [float(H0)]]))
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    # one max-reduce serves both ends of the range: (-qmin, qmax)
    mm = torch.tensor([-float(qmin), float(qmax)], dtype=torch.float64)
    dist.all_reduce(mm, op=dist.ReduceOp.MAX, group=group)
    out = buf.numpy()
    return out[:N * N].reshape(N, N).copy(), out[N * N:N * N + N].copy(), float(out[-1]), -float(mm[0]), float(mm[1])


class HostComm:
    """The same reduction as RcclComm.allreduce_stats, staged through the host over any torch.distributed backend (gloo): the
    packed statistics come off the device (fh_stats_get_packed, ~380 KB at N = 300), are summed over the ranks of the default
    process group, and go back (fh_stats_set_packed).  For ranks that SHARE a device -- RCCL refuses two ranks on one GPU --
    and for boxes without RCCL; FRANK_AMD_COMM=host selects it in make_comm."""

    def __init__(self, rank, world, device=0, broadcast_bytes=None, group=None):
        from frank_amd import _lib
        self._lib, self._world, self._group = _lib, int(world), group
        self.handle = True
        self._ms = 0.0

    def allreduce_stats(self, ctx):
        import ctypes
        import time
        import torch
        import torch.distributed as dist
        L = self._lib
        n = ctypes.c_int64(0)
        L.check(L.lib.fh_stats_device(ctx, None, ctypes.byref(n), None))
        buf, mm = np.empty(n.value), np.empty(2)
        L.check(L.lib.fh_stats_get_packed(ctx, L.ptr(buf), n.value, L.ptr(mm)))
        t0 = time.perf_counter()
        tb = torch.from_numpy(buf)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM, group=self._group)
        # (-qmin, qmax): NaN is the device's neutral element "nothing binned"; a max over ranks wants -inf instead
        tm = torch.from_numpy(np.where(np.isnan(mm), -np.inf, mm))
        dist.all_reduce(tm, op=dist.ReduceOp.MAX, group=self._group)
        self._ms = 1e3 * (time.perf_counter() - t0)
        mm = tm.numpy()
        mm = np.ascontiguousarray(np.where(np.isinf(mm) & (mm < 0), np.nan, mm))
        L.check(L.lib.fh_stats_set_packed(ctx, L.ptr(buf), n.value, L.ptr(mm)))

    def last_allreduce_ms(self):
        return self._ms

    def size(self):
        return self._world

    def close(self):
        self.handle = None


def make_comm(rank, world, device, broadcast_bytes, group=None):
    """RcclComm, or HostComm when FRANK_AMD_COMM=host (ranks sharing a device, no RCCL)."""
    import os
    if os.environ.get("FRANK_AMD_COMM", "rccl").lower() == "host":
        return HostComm(rank, world, device, broadcast_bytes, group)
    return RcclComm(rank, world, device, broadcast_bytes)


class RcclComm:
    """RCCL communicator for the device-resident statistics (one rank per GPU)."""

    def __init__(self, rank, world, device, broadcast_bytes):
        """`broadcast_bytes(b: bytes | None) -> bytes` ships rank 0's 128-byte unique id to every rank."""
        import ctypes
        from frank_amd import _lib
        self._lib = _lib
        ident = None
        if rank == 0:
            buf = ctypes.create_string_buffer(128)
            _lib.check(_lib.lib.fh_comm_unique_id(buf))
            ident = buf.raw
        ident = broadcast_bytes(ident)
        self.handle = ctypes.c_void_p()
        _lib.check(_lib.lib.fh_comm_create(ident, rank, world, device, ctypes.byref(self.handle)))

    def allreduce_stats(self, ctx):
        """Sum the context's device-resident statistics over the ranks (asynchronous on the context's stream)."""
        self._lib.check(self._lib.lib.fh_comm_allreduce_stats(self.handle, ctx))

    def last_allreduce_ms(self):
        """Device time of the most recent allreduce_stats (HIP events on the context's stream)."""
        import ctypes
        ms = ctypes.c_float(0)
        self._lib.check(self._lib.lib.fh_comm_last_allreduce_ms(self.handle, ctypes.byref(ms)))
        return ms.value

    def size(self):
        return self._lib.lib.fh_comm_size(self.handle)

    def close(self):
        if self.handle:
            self._lib.lib.fh_comm_destroy(self.handle)
            self.handle = None

"""world_size-2 gloo test of the sharded mapping (CPU): each rank bins its slab (the oracle is the compute
checker here), the statistics are all-reduced, and every rank must hold the unsharded result."""
import os
import socket

import numpy as np
import pytest

from conftest import rel_to_max
from frank_amd.constants import rad_to_arcsec
from frank_amd.distributed import allreduce_mapping, shard_range
from frank_amd.mock import MOCK_GEOMETRY, mock_disc_visibilities

GEOM = (MOCK_GEOMETRY["inc"], MOCK_GEOMETRY["PA"], MOCK_GEOMETRY["dRA"], MOCK_GEOMETRY["dDec"])
RMAX = 2.0 / rad_to_arcsec
N, NVIS = 40, 6001


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 6001, 10 ** 7):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from oracle import oracle as fo
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    u, v, V, w = mock_disc_visibilities(NVIS, seed=3, noise_seed=4)
    first, count = shard_range(NVIS, rank, world)
    sl = slice(first, first + count)
    m = fo.map_visibilities(N, RMAX, GEOM, u[sl], v[sl], V[sl], w[sl], check_qbounds=False)
    M, j, H0, qmin, qmax = allreduce_mapping(m["M"], m["j"], m["null_likelihood"], m["qmin"], m["qmax"])
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), M=M, j=j, H0=H0, qmin=qmin, qmax=qmax)
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce(tmp_path):
    import torch.multiprocessing as mp
    from oracle import oracle as fo
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    u, v, V, w = mock_disc_visibilities(NVIS, seed=3, noise_seed=4)
    full = fo.map_visibilities(N, RMAX, GEOM, u, v, V, w, check_qbounds=False)
    for r in range(2):
        g = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert rel_to_max(g["M"], full["M"]) < 1e-13
        assert rel_to_max(g["j"], full["j"]) < 1e-13
        assert abs(float(g["H0"]) - full["null_likelihood"]) < 1e-11 * abs(full["null_likelihood"])
        assert float(g["qmin"]) == full["qmin"] and float(g["qmax"]) == full["qmax"]
    a, b = (np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(2))
    assert np.array_equal(a["M"], b["M"]) and np.array_equal(a["j"], b["j"])

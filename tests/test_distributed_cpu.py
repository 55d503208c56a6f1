"""N > 1 path on CPU: world_size-2 gloo processes shard the batch, run the local filter (the
oracle's C port stands in for the HIP kernel -- this test is about sharding and the collective,
the GPU parity tests cover the kernel) and all-gather the posterior summaries."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import common as cm


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, T, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bayesianfiltering_amd import distributed as bd
    from oracle import c_oracle
    a = cm.cv_model_arrays()
    ys = cm.simulate_batch(a, B, T, seed=123)          # every rank builds the same global batch

    def local_filter(y_block):
        init = np.tile(a["m0"], (y_block.shape[0], 1))
        return c_oracle.kalman_filter(a, y_block, init, fields=("means", "covariances"))

    def summary(res):                                   # final mean and covariance per trajectory
        m = torch.from_numpy(res["means"][:, 0, -1])
        P = torch.from_numpy(res["covariances"][:, 0, -1].reshape(m.shape[0], -1))
        return torch.cat([m, P], dim=1)

    res, gathered = bd.filter_sharded(local_filter, ys, summary=summary)
    lo, hi = bd.shard_bounds(B, rank, world)
    assert res["means"].shape[0] == hi - lo
    if rank == 0:
        np.save(out_path, gathered.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 7])
def test_two_rank_gloo_sharding_and_allgather(tmp_path, B):
    T = 24
    out_path = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), B, T, out_path), nprocs=2, join=True)
    from oracle import c_oracle
    a = cm.cv_model_arrays()
    ys = cm.simulate_batch(a, B, T, seed=123)
    ref = c_oracle.kalman_filter(a, ys, np.tile(a["m0"], (B, 1)), fields=("means", "covariances"))
    exp = np.concatenate([ref["means"][:, 0, -1], ref["covariances"][:, 0, -1].reshape(B, -1)], axis=1)
    got = np.load(out_path)
    assert got.shape == (B, 20)
    assert np.array_equal(got, exp)                     # same arithmetic on every shard: bit-exact


def test_shard_bounds_cover_the_batch():
    from bayesianfiltering_amd.distributed import shard_bounds
    for B in (1, 7, 8, 65536, 65537):
        for world in (1, 2, 4, 8):
            blocks = [shard_bounds(B, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == B
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1

"""The N > 1 path around the HIP kernels on the GPU box: `distributed.filter_sharded` shards the batch over the ranks of
a torch.distributed group, every rank runs the ENGINE (bf_kalman_filter_f32 / bf_bpf_f32 on cuda:0) on its block, and the
per-trajectory summaries are all-gathered.  One GPU is available to the tests, so two ranks share it and the collective
runs over gloo on host copies of the summaries (RCCL refuses two ranks on one device); the shard arithmetic, the kernel
launches under an initialised process group and the gather are what `bench.py --gpus N` does on N GPUs."""
import os
import socket

import numpy as np
import pytest

from tests import common as cm

pytestmark = pytest.mark.gpu
F32 = np.float32


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, T, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)       # the process group first, then the GPU
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import distributed as bd
    a = cm.cv_model_arrays()
    p = cm.product_params(a)
    ys = torch.as_tensor(cm.simulate_batch(a, B, T, seed=123))          # every rank builds the same global batch

    def local_filter(y_block):
        return bfa.kalman_filter(p, y_block.cuda(), initial_means=np.tile(a["m0"], (y_block.shape[0], 1)))

    def summary(post):                                                   # final mean and covariance per trajectory
        return torch.cat([post.means[:, 0, -1], post.covariances[:, 0, -1].reshape(post.means.shape[0], -1)], dim=1).cpu()

    post, gathered = bd.filter_sharded(local_filter, ys, summary=summary)
    lo, hi = bd.shard_bounds(B, rank, world)
    assert post.means.shape[0] == hi - lo and post.means.is_cuda
    if rank == 0:
        np.save(out_path, gathered.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [70, 131])
def test_two_ranks_shard_the_hip_kernel(tmp_path, B):
    import torch.multiprocessing as mp
    import bayesianfiltering_amd as bfa
    T = 40
    out_path = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), B, T, out_path), nprocs=2, join=True)
    a = cm.cv_model_arrays()
    ys = cm.simulate_batch(a, B, T, seed=123)
    init = np.tile(a["m0"], (B, 1))
    whole = bfa.kalman_filter(cm.product_params(a), ys, initial_means=init)
    exp = np.concatenate([whole.means[:, 0, -1].cpu().numpy(), whole.covariances[:, 0, -1].reshape(B, -1).cpu().numpy()], axis=1)
    got = np.load(out_path)
    assert got.shape == (B, 20)
    assert np.array_equal(got, exp)                     # sharded == unsharded, bit for bit (trajectories are independent)
    ref = cm.oracle_kalman_batch(a, ys, init)
    assert cm.rel_err(got[:, :4], ref["means"][:, 0, -1]) < 1e-5


def test_filter_sharded_single_rank_is_the_plain_call():
    """World size 1 (no process group): filter_sharded is the local filter plus an identity gather."""
    import torch
    import bayesianfiltering_amd as bfa
    from bayesianfiltering_amd import distributed as bd
    a = cm.cv_model_arrays()
    ys = torch.as_tensor(cm.simulate_batch(a, 33, 20, seed=5), device="cuda")
    p = cm.product_params(a)
    post, gathered = bd.filter_sharded(lambda y: bfa.kalman_filter(p, y), ys, summary=lambda r: r.means[:, 0, -1])
    plain = bfa.kalman_filter(p, ys)
    assert torch.equal(gathered, plain.means[:, 0, -1]) and torch.equal(post.covariances, plain.covariances)


def test_c_abi_allgather_over_an_rccl_communicator():
    """bf_allgather_summaries on a communicator this test creates itself with ncclCommInitAll over the one
    visible GPU -- world size 1, so the gather is a device copy, but the call goes through RCCL's ncclAllGather on the
    caller's stream exactly as it does on eight ranks."""
    import ctypes as C
    import torch
    from bayesianfiltering_amd import _lib
    lib = _lib.require_gpu()
    rccl = None
    for cand in (os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "librccl.so", "/opt/rocm/lib/librccl.so"):
        try:
            rccl = C.CDLL(cand, mode=C.RTLD_GLOBAL)
            break
        except OSError:
            continue
    assert rccl is not None
    comm = C.c_void_p()
    dev = (C.c_int * 1)(0)
    assert rccl.ncclCommInitAll(C.byref(comm), 1, dev) == 0
    try:
        send = torch.arange(1000, dtype=torch.float32, device="cuda") * 0.5
        recv = torch.zeros_like(send)
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(lib.bf_allgather_summaries(C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), send.numel() * 4, comm,
                                              C.c_void_p(stream)))
        torch.cuda.synchronize()
        assert torch.equal(send, recv)
        assert lib.bf_allgather_summaries(None, C.c_void_p(recv.data_ptr()), 4, comm, None) == _lib.BF_EINVAL
    finally:
        rccl.ncclCommDestroy(comm)

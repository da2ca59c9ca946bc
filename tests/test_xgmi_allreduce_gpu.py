"""XgmiAllReduce (sglk_allreduce_sum_bf16): the direct peer-memory all-reduce that stands where the reference has its
shared-memory all-reduce between the ranks of one host (/root/reference/test_allreduce.py:86-105: bf16 SUM, in place, checked
against the sum of the per-rank tensors).

A 1-GPU box cannot show xGMI, but it can show everything else: here 2 and 4 processes share cuda:0, exchange HIP IPC handles
through a gloo group and reduce through each other's mapped staging regions -- the same mapping, flags, epochs, parities and
kernels as one process per GPU.  Expected bits: fp32 sum in ascending rank order, rounded to bf16 once.
"""
import os
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = [8, 5120, 40 * 5120 + 8, 1024 * 5120]            # the last one is the reference's bench message (10 MiB of bf16)


def _inputs(world, n, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(world, n, generator=g) * 3).bfloat16()


def _expected(full):
    acc = full[0].float()
    for r in range(1, full.shape[0]):
        acc = acc + full[r].float()
    return acc.bfloat16()


def _worker(rank, world, port, ret, stall):
    for p in (ROOT, os.path.join(ROOT, "sgl-cpu-tests_amd")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = None
    try:
        import sgl_kernel  # noqa: F401
        from sgl_kernel import collectives as ops
        torch.cuda.set_device(0)
        comm = ops.XgmiAllReduce(None, max_bytes=16 << 20)
        checks = {}
        for algo in (1, 2, 0):
            for n in SIZES:
                for rep in range(3):                      # consecutive calls: both parities of the staging region, epochs grow
                    full = _inputs(world, n, 1000 * algo + n % 997 + rep)
                    t = full[rank].cuda()
                    comm.all_reduce(t, algo)
                    comm.check()
                    checks[(algo, n, rep)] = torch.equal(t.cpu(), _expected(full))
        # through the reference's entry point: the registered communicator takes bf16 sums, the group (gloo, host staged) the rest
        full = _inputs(world, 5120, 5)
        t = full[rank].cuda()
        e0 = comm.epoch
        assert ops.shm_allreduce(t, None, dist.ReduceOp.SUM) is None
        comm.check()
        checks["shm_allreduce"] = torch.equal(t.cpu(), _expected(full)) and comm.epoch == e0 + 1
        f32 = full[rank].float().cuda()
        ops.shm_allreduce(f32, None, dist.ReduceOp.SUM)
        checks["fallback_f32"] = comm.epoch == e0 + 1 and torch.allclose(f32.cpu(), full.float().sum(0), rtol=1e-5, atol=1e-5)
        # back-to-back calls without a host sync in between (what a decode loop does), timed
        t = _inputs(world, SIZES[-1], 9)[rank].cuda()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            comm.all_reduce(t.clone())
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        comm.check()
        dist.barrier()
        timed_out = None
        if stall:
            # a peer that never arrives: the kernels give up after SGLK_AR_WAIT_MS and the status word says so (no hang); the NEXT
            # call reports it too, unasked (host mirror of the status word); resync() on every rank makes the communicator usable
            # again
            from sgl_kernel import _lib
            os.environ["SGLK_AR_WAIT_MS"] = "1500"
            _lib.lib().sglk_reload_env()
            if rank == 0:
                x = torch.ones(8, dtype=torch.bfloat16, device="cuda")
                comm.all_reduce(x)
                try:
                    comm.check()
                    timed_out = False
                except RuntimeError:
                    timed_out = True
                try:
                    comm.all_reduce(x)
                    timed_out = False
                except RuntimeError:
                    pass
            dist.barrier()
            comm.resync()
            full = _inputs(world, 4096, 77)
            t = full[rank].cuda()
            comm.all_reduce(t)
            comm.check()
            checks["after_resync"] = torch.equal(t.cpu(), _expected(full))
            # a misaligned view on ONE rank: every rank still takes the direct path (the choice is rank-invariant), that rank stages
            base = torch.zeros(4096 + 8, dtype=torch.bfloat16, device="cuda")
            v = base[4:4 + 4096] if rank == 0 else base[:4096]
            v.copy_(full[rank])
            assert (v.data_ptr() % 16 != 0) == (rank == 0)
            e0 = comm.epoch
            ops.shm_allreduce(v, None, dist.ReduceOp.SUM)
            comm.check()
            checks["misaligned_on_one_rank"] = torch.equal(v.cpu(), _expected(full)) and comm.epoch == e0 + 1
        ret[rank] = (checks, ms, timed_out)
    finally:
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


@pytest.mark.timeout(420)
@pytest.mark.parametrize("world,stall", [(2, True), (4, False)])
def test_allreduce_between_processes_sharing_the_gpu(world, stall):
    port = 33500 + (os.getpid() % 2000) + world
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, ret, stall), nprocs=world, join=True)
        for rank in range(world):
            checks, ms, timed_out = ret[rank]
            bad = [k for k, v in checks.items() if not v]
            assert not bad, (rank, bad)
            if stall and rank == 0:
                assert timed_out is True, "a missing peer must end in the status word, not in a hang"
        print(f"\n[xgmi allreduce, {world} processes on one GPU] 10 MiB bf16: {ret[0][1]:.3f} ms per call")

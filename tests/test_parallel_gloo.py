"""CPU, world_size 2 over gloo: the gradient all-reduce bookkeeping of the data-parallel loop.
W ranks x (nAveGrad/W) micro-batches must equal the single-process accumulated gradient."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from osvos_pytorch_amd.parallel import GradientAllReducer, shard_indices
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 1, 1))
    frozen = torch.nn.Parameter(torch.ones(3))          # a parameter that never gets a gradient
    model.register_parameter("frozen", frozen)
    red = GradientAllReducer(model, average=False)
    red.broadcast_parameters(0)
    data = torch.randn(4, 1, 3, 8, 8, generator=torch.Generator().manual_seed(7))
    n_ave = 4
    for i in shard_indices(4, rank, world):
        (model(data[i]).sum() / n_ave).backward()
    red.all_reduce()
    flat = torch.cat([p.grad.flatten() for p in model.parameters() if p.grad is not None])
    if rank == 0:
        torch.save(flat, out)
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_single_process_accumulation(tmp_path):
    out = str(tmp_path / "g.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 1, 1))
    data = torch.randn(4, 1, 3, 8, 8, generator=torch.Generator().manual_seed(7))
    for i in range(4):
        (model(data[i]).sum() / 4).backward()
    ref = torch.cat([p.grad.flatten() for p in model.parameters()])
    torch.testing.assert_close(got, ref, rtol=1e-6, atol=1e-7)


def test_shard_indices_partition():
    from osvos_pytorch_amd.parallel import shard_indices
    for n, w in [(20, 8), (5, 2), (3, 4)]:
        parts = [shard_indices(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))

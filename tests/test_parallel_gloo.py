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


def test_flat_gradient_arena_views_zero_and_reattach():
    """the reducer keeps every .grad as a view into one flat buffer: accumulation lands in it, zero_grads is one memset that
    keeps the views, a foreign optimizer.zero_grad() (set_to_none) is repaired at the next call, parameters without a
    gradient stay out of the buffer"""
    sys.path.insert(0, REPO)
    from osvos_pytorch_amd.parallel import GradientAllReducer
    torch.manual_seed(1)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(4, 1, 1))
    model.register_parameter("frozen", torch.nn.Parameter(torch.ones(3)))
    red = GradientAllReducer(model)
    assert red.attach() is None
    red.zero_grads()                                   # before the first backward: plain None
    x = torch.randn(2, 3, 8, 8)
    model(x).sum().backward()
    ref = [p.grad.clone() for p in model.parameters() if p.grad is not None]
    flat = red.attach()
    live = [p for p in model.parameters() if p.grad is not None]
    assert flat.numel() == sum(p.numel() for p in live) and model.frozen.grad is None
    assert all(p.grad.data_ptr() >= flat.data_ptr() and p.grad.data_ptr() < flat.data_ptr() + flat.numel() * 4 for p in live)
    assert all(torch.equal(p.grad, r) for p, r in zip(live, ref))
    model(x).sum().backward()                          # autograd accumulates INTO the views
    assert torch.allclose(flat, torch.cat([2 * r.flatten() for r in ref]))
    ptrs = [p.grad.data_ptr() for p in live]
    red.zero_grads()
    assert float(flat.abs().max()) == 0.0 and [p.grad.data_ptr() for p in live] == ptrs
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    model(x).sum().backward()
    opt.zero_grad()                                    # drops the views (set_to_none)
    model(x).sum().backward()
    assert red.attach() is flat and [p.grad.data_ptr() for p in live] == ptrs
    assert torch.allclose(flat, torch.cat([r.flatten() for r in ref]))
    red.all_reduce()                                   # no process group: a no-op that leaves the arena attached
    assert [p.grad.data_ptr() for p in live] == ptrs


def _count_worker(rank, world, port, out):
    sys.path.insert(0, REPO)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from osvos_pytorch_amd.parallel import GradientAllReducer, cbce_with_counts, global_class_counts
    g = torch.Generator().manual_seed(11)
    logits = torch.randn(4, 1, 12, 16, generator=g, dtype=torch.float64) * 3
    labels = (torch.rand(4, 1, 12, 16, generator=g) > 0.7).float()
    labels[3] = 0                                        # one frame without positives: the counts must come from the whole batch
    mine = slice(2 * rank, 2 * rank + 2)                 # this rank's half of the batch
    x = logits[mine].clone().requires_grad_()
    n_pos, n_tot, n_img = global_class_counts(labels[mine])
    loss = cbce_with_counts(x, labels[mine], n_pos, n_tot, n_img)
    loss.backward()
    t = torch.stack([loss.detach()])
    dist.all_reduce(t)
    # broadcast_parameters: ONE flat collective
    torch.manual_seed(100 + rank)                        # different weights per rank before the broadcast
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Conv2d(4, 2, 1))
    GradientAllReducer(model).broadcast_parameters(0)
    torch.save({"loss": t[0], "grad": x.grad, "counts": (float(n_pos), float(n_tot), float(n_img)),
                "w": torch.cat([p.data.flatten() for p in model.parameters()])}, out + ".%d" % rank)
    dist.destroy_process_group()


def test_count_exchange_makes_a_sharded_batch_equal_the_single_process_batch_loss(tmp_path):
    """SURVEY 8e row 3 / reference osvos_layers.py:28-34,46: the class weights are counted over the WHOLE batch tensor and the loss is divided
    by the batch size -- two ranks holding half a batch each must exchange (n_pos, n_total, N) to reproduce it.  Also: broadcast_parameters
    as one flat collective leaves every rank with rank 0's weights."""
    sys.path.insert(0, REPO)
    from oracle import torch_ref
    out = str(tmp_path / "c")
    port = 30500 + (os.getpid() % 2000)
    mp.spawn(_count_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    g = torch.Generator().manual_seed(11)
    logits = torch.randn(4, 1, 12, 16, generator=g, dtype=torch.float64) * 3
    labels = (torch.rand(4, 1, 12, 16, generator=g) > 0.7).float()
    labels[3] = 0
    x = logits.clone().requires_grad_()
    ref = torch_ref.cbce_loss(x, labels, size_average=False)          # the reference formula on the whole batch
    ref.backward()
    assert r0["counts"] == r1["counts"] == (float((labels >= 0.5).sum()), float(labels.numel()), 4.0)
    torch.testing.assert_close(r0["loss"], ref.detach(), rtol=1e-12, atol=0)
    torch.testing.assert_close(torch.cat([r0["grad"], r1["grad"]]), x.grad, rtol=1e-12, atol=1e-15)
    assert torch.equal(r0["w"], r1["w"])
    torch.manual_seed(100)
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Conv2d(4, 2, 1))
    assert torch.equal(r0["w"], torch.cat([p.data.flatten() for p in model.parameters()]))


def test_abi_communicator_reports_a_taken_port_instead_of_hanging(monkeypatch):
    """the C-ABI communicator's id store used MASTER_PORT + 1 silently -- a second job on the node collided.  The port is
    OSVOS_COMM_PORT now, the rendezvous is opened before RCCL or the device are touched, and a taken port raises with the variable's name."""
    import socket
    from osvos_pytorch_amd import parallel
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    s.listen(1)
    port = s.getsockname()[1]
    monkeypatch.setenv("OSVOS_COMM_PORT", str(port))
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("OSVOS_COMM_TIMEOUT", "5")
    assert parallel.comm_port() == port
    try:
        with pytest.raises(RuntimeError, match="OSVOS_COMM_PORT"):
            parallel.AbiCommunicator(0, 2, "cpu")
    finally:
        s.close()
    monkeypatch.delenv("OSVOS_COMM_PORT")
    monkeypatch.setenv("MASTER_PORT", "29500")
    assert parallel.comm_port() == 29501
    monkeypatch.setenv("OSVOS_COMM_PORT", "80")
    with pytest.raises(ValueError):
        parallel.comm_port()

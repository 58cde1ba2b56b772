#!/usr/bin/env python3
"""Single-GPU check of the data-parallel path on the real backend: an 'nccl' (= RCCL) process group with ONE rank, the gradient
all-reduce forced on (GradientAllReducer(always=True)), the parent TrainLoop on a small synthetic set -- against the same loop without
any process group.  A one-rank sum is the identity, so parameters, momentum-free of noise, must come out BIT-IDENTICAL; the run also
has to take the overlapped path (seven chunked collectives on the communication stream behind the gradient-ready events) from the
second optimizer step on.  Prints one line per check; exit code 0 = all good.  (tests/test_gpu_scripts.py runs it; the log is kept
under profiles/.)"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29571")
os.environ.setdefault("OSVOS_DP_OVERLAP", "1")          # the check exercises the overlapped path (opt-in in production)

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import networks.vgg_osvos as vo  # noqa: E402
from osvos_pytorch_amd.parallel import GradientAllReducer  # noqa: E402
from osvos_pytorch_amd.train_common import TrainLoop, epoch_plan, make_sgd  # noqa: E402


def frames(n, h, w, dev):
    out = []
    for i in range(n):
        g = torch.Generator().manual_seed(300 + i)
        img = torch.randn(1, 3, h, w, generator=g) * 40.0
        yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        gt = ((((yy - 0.5 * h) / (0.3 * h)) ** 2 + ((xx - (0.3 + 0.05 * i) * w) / (0.2 * w)) ** 2) <= 1).float()[None, None]
        out.append((img.to(dev), gt.to(dev)))
    return out


def run(dev, use_reducer, n_ave, epochs, data, sd0, overlap=True, comm=None):
    sys.stdout, keep = open(os.devnull, "w"), sys.stdout
    try:
        net = vo.OSVOS(pretrained=0)
    finally:
        sys.stdout.close()
        sys.stdout = keep
    net.load_state_dict(sd0)
    net.to(dev)
    opt = make_sgd(net, "parent", lr=1e-8)
    red = GradientAllReducer(net, average=False, always=True, overlap=overlap, comm=comm) if use_reducer else None
    loop = TrainLoop(net, opt, mode="parent", n_ave_grad=n_ave, n_epochs=8, reducer=red)
    t0 = time.perf_counter()
    for epoch in range(epochs):
        for idx, _ in epoch_plan(len(data), epoch, n_ave, 0, 1, seed=3):
            loop.micro_batch(data[idx][0].clone().requires_grad_(), data[idx][1], epoch=epoch)
    torch.cuda.synchronize()
    return net, loop, red, time.perf_counter() - t0


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    # both communicators come up BEFORE anything else touches the device: an RCCL communicator initialised after the first allocations /
    # launches leaves every later step ~30 % slower (profiles/r03_dp_backends.txt) -- the timing lines below would measure that instead
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    dist.barrier()
    from osvos_pytorch_amd.parallel import AbiCommunicator
    comm = AbiCommunicator(0, 1, dev)                      # RCCL through the C ABI (osvos_comm_*), one rank
    h, w = int(os.environ.get("DP_H", "120")), int(os.environ.get("DP_W", "214"))
    data = frames(6, h, w, dev)
    from bench import synth_problem               # He-init weights with calibrated heads (logit maps ~ N(-1, 3^2)): finite, well-scaled gradients
    ref_net, _, _ = synth_problem(1, h, w, dev, seed=0)
    sd0 = {k: v.detach().cpu().clone() for k, v in ref_net.state_dict().items()}
    del ref_net
    plain, loop_p, _, t_plain = run(dev, False, 3, 3, data, sd0)
    if os.environ.get("DP_TIME", "0") == "1":
        run(dev, False, 5, 2, data, sd0)
        t0 = time.perf_counter()
        run(dev, False, 5, 10, data, sd0)
        print("timing, no reducer: %.3f ms per micro-batch (60 micro-batches, nAveGrad 5, incl. building the net)" % ((time.perf_counter() - t0) / 60 * 1e3))
    blocking, loop_b, red_b, t_block = run(dev, True, 3, 3, data, sd0, overlap=False)
    forced, loop_f, red, t_forced = run(dev, True, 3, 3, data, sd0)
    ok = True

    def report(tag, a, b):
        worst = sorted(((float((x.double() - y.double()).abs().max() / (y.double().abs().max() + 1e-30)), k)
                        for (k, x), y in zip(a.state_dict().items(), b.state_dict().values())), reverse=True)
        print("%s: %d of %d tensors differ; largest relative difference %.3g (%s)" % (tag, sum(1 for e, _ in worst if e > 0), len(worst), worst[0][0], worst[0][1]))
    nan = sum(int(torch.isnan(v).any()) for v in plain.state_dict().values())
    print("tensors with NaN after training (must be 0): %d" % nan)
    ok &= nan == 0
    report("blocking one-rank all-reduce vs no process group", blocking, plain)
    report("overlapped chunked all-reduce vs no process group", forced, plain)
    report("overlapped vs blocking", forced, blocking)
    abi_b, loop_ab, red_ab, t_ab = run(dev, True, 3, 3, data, sd0, overlap=False, comm=comm)
    abi_o, loop_ao, red_ao, t_ao = run(dev, True, 3, 3, data, sd0, overlap=True, comm=comm)
    report("C-ABI RCCL, blocking all-reduce vs no process group", abi_b, plain)
    report("C-ABI RCCL, overlapped chunked all-reduce vs no process group", abi_o, plain)
    same_abi = all(torch.equal(a, b) for a, b in zip(plain.state_dict().values(), abi_b.state_dict().values())) and \
        all(torch.equal(a, b) for a, b in zip(plain.state_dict().values(), abi_o.state_dict().values()))
    print("C-ABI RCCL paths: %s; overlapped steps %d of %d" % ("bit-identical" if same_abi else "DIFFERENT", red_ao.overlapped_steps, loop_ao.steps))
    ok &= same_abi and red_ao.overlapped_steps >= loop_ao.steps - 1 and red_ab.overlapped_steps == 0
    print("wall time with the blocking all-reduce: %.3f s (%d overlapped steps)" % (t_block, red_b.overlapped_steps))
    same = all(torch.equal(a, b) for a, b in zip(plain.state_dict().values(), forced.state_dict().values()))
    print("parameters after %d optimizer steps, RCCL one-rank all-reduce forced vs no process group: %s" % (loop_f.steps, "bit-identical" if same else "DIFFERENT"))
    ok &= same and loop_f.steps == loop_p.steps == 6
    moved = sum(int(not torch.equal(sd0[k].to(dev), v)) for k, v in forced.state_dict().items())
    print("tensors changed by training: %d of %d" % (moved, len(sd0)))
    ok &= moved >= 25
    print("optimizer steps reduced on the overlapped path (chunked collectives behind gradient-ready events): %d of %d" % (red.overlapped_steps, loop_f.steps))
    ok &= red.overlapped_steps >= loop_f.steps - 1
    print("chunks of the flat gradient arena (group, first, last element):", red._slices)
    ok &= len(red._slices) == 7 and red._slices[0][0] == 0 and red._slices[-1][0] == 6
    print("wall time of the loop: %.3f s without a process group, %.3f s with the forced one-rank all-reduce (%dx%d, 18 micro-batches)" % (t_plain, t_forced, w, h))
    if os.environ.get("DP_TIME", "0") == "1":      # per-micro-batch cost of the three variants at this size (process group still up)
        big = frames(6, h, w, dev)
        for tag, kw in (("no reducer (communicators up)", dict(use_reducer=False)), ("torch.distributed, blocking all-reduce", dict(use_reducer=True, overlap=False)),
                        ("torch.distributed, overlapped chunked all-reduce", dict(use_reducer=True, overlap=True)),
                        ("C-ABI RCCL, blocking all-reduce", dict(use_reducer=True, overlap=False, comm=comm)),
                        ("C-ABI RCCL, overlapped chunked all-reduce", dict(use_reducer=True, overlap=True, comm=comm))):
            _, lp, _, _ = run(dev, kw["use_reducer"], 5, 2, big, sd0, overlap=kw.get("overlap", True), comm=kw.get("comm"))       # warm
            t0 = time.perf_counter()
            _, lp, _, _ = run(dev, kw["use_reducer"], 5, 10, big, sd0, overlap=kw.get("overlap", True), comm=kw.get("comm"))
            dt = time.perf_counter() - t0
            print("timing, %s: %.3f ms per micro-batch (60 micro-batches, nAveGrad 5, incl. building the net)" % (tag, dt / 60 * 1e3))
    comm.close()
    dist.destroy_process_group()
    print("DP_SELFCHECK_OK" if ok else "DP_SELFCHECK_FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

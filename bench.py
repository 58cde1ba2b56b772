#!/usr/bin/env python3
"""bench.py -- headline benchmark of the OSVOS hot path on MI355X.

Metric (BASELINE.json): frames/sec (fwd+bwd) of OSVOS-VGG16 at 854x480 per GPU.
Workload at N=1 (BASELINE.json configs[1]): the inner loop of train_online.py:112-149 restated --
batch 1, 854x480, fused-head class-balanced BCE, loss/nAveGrad, backward, SGD step (8 param groups of
train_online.py:79-88) every nAveGrad=5 iterations -- on a seeded synthetic frame already resident
in HBM (no dataloader on either side), fp32.  One "step" = one forward+loss+backward of one frame
(including, every 5th step, the optimizer step).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N>1: one process per GPU, every rank fine-tunes on its own frames (weak scaling) and the flat
gradient buffer is all-reduced (RCCL over xGMI via torch.distributed 'nccl') once per optimizer step.

Prints ONE JSON line on rank 0 with the contract fields plus `roofline` (dominant kernel family:
the conv3x3 implicit-GEMM MFMA kernel; achieved = algorithmic FLOPs / summed launch durations
measured with HIP events on the launch stream inside the timed region) and `cpu_baseline` (the
torch-CPU restatement of the reference loop from oracle/torch_ref.py on this node's host cores).
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

# ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The backward uses three streams (data gradients /
# weight gradients / slab reduces); once torch.distributed's RCCL communicator adds its own streams two of ours end up on the same
# hardware queue and serialise -- measured -7 % (128 -> 119 frames/s) from init_process_group alone.  Eight queues restore it.
# Must be set before the HIP runtime initialises, i.e. before the first CUDA call of the process.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense (no 2:1 sparsity)
# algorithmic FLOPs per 854x480 frame, SURVEY.md 8(d): 3x3 convs only, fwd + dgrad + wgrad
GFLOP_FWD_480P = 258.229


def conv_gflop_forward(h, w):
    """2*H*W*Cout*9*Cin summed over the 13 trunk + 4 side_prep convs."""
    chans = [[64, 64], [128, 128], [256, 256, 256], [512, 512, 512], [512, 512, 512]]
    total, cin = 0.0, 3
    for si, st in enumerate(chans):
        if si > 0:
            h, w = (h + 1) // 2, (w + 1) // 2
        for c in st:
            total += 2.0 * h * w * c * 9 * cin
            cin = c
        if si > 0:
            total += 2.0 * h * w * 16 * 9 * cin
    return total / 1e9


def synth_problem(n, h, w, device, seed):
    """Seeded synthetic frame + mask + He-init weights with calibrated heads, built
    on the product path itself (the oracle is not imported here)."""
    import networks.vgg_osvos as vo
    g = torch.Generator().manual_seed(1234 + seed)
    coarse = torch.randn(n, 3, max(2, h // 16 + 2), max(2, w // 16 + 2), generator=g)
    x = torch.nn.functional.interpolate(coarse, size=(h, w), mode="bilinear", align_corners=True) * 50.0
    x = x + torch.randn(n, 3, h, w, generator=g) * 8.0
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    m = ((((yy - 0.5 * h) / (0.27 * h)) ** 2 + ((xx - 0.45 * w) / (0.24 * w)) ** 2) <= 1.0).float()
    m = m[None, None].repeat(n, 1, 1, 1)
    sys.stdout, keep = open(os.devnull, "w"), sys.stdout
    try:
        net = vo.OSVOS(pretrained=0)
    finally:
        sys.stdout.close()
        sys.stdout = keep
    for mod in net.modules():
        if isinstance(mod, torch.nn.Conv2d):
            std = math.sqrt(2.0 / (mod.kernel_size[0] * mod.kernel_size[1] * mod.out_channels))
            mod.weight.data.copy_(torch.randn(mod.weight.shape, generator=g) * std)
            mod.bias.data.zero_()
    net = net.to(device)
    x, m = x.to(device), m.to(device)
    with torch.no_grad():      # head calibration: every logit map ~ N(-1, 3^2)
        outs = net.forward(x)
        for i in range(4):
            s = 3.0 / float(outs[i].std())
            net.score_dsn[i].weight.mul_(s)
            net.score_dsn[i].bias.fill_(-1.0 - float(outs[i].mean()) * s)
        s = 3.0 / float(outs[4].std())
        net.fuse.weight.mul_(s)
        net.fuse.bias.fill_(-1.0 - float(outs[4].mean()) * s)
    return net, x, m


def make_optimizer(net, mode):
    """train_online.py:79-88 / train_parent.py:87-103."""
    lr, wd = 1e-8, 0.0002
    groups = [
        {'params': [p for n, p in net.stages.named_parameters() if 'weight' in n], 'weight_decay': wd},
        {'params': [p for n, p in net.stages.named_parameters() if 'bias' in n], 'lr': lr * 2},
        {'params': [p for n, p in net.side_prep.named_parameters() if 'weight' in n], 'weight_decay': wd},
        {'params': [p for n, p in net.side_prep.named_parameters() if 'bias' in n], 'lr': lr * 2},
    ]
    if mode == "parent":
        groups += [
            {'params': [p for n, p in net.score_dsn.named_parameters() if 'weight' in n], 'lr': lr / 10, 'weight_decay': wd},
            {'params': [p for n, p in net.score_dsn.named_parameters() if 'bias' in n], 'lr': 2 * lr / 10},
        ]
    groups += [
        {'params': [p for n, p in net.upscale.named_parameters() if 'weight' in n], 'lr': 0},
        {'params': [p for n, p in net.upscale_.named_parameters() if 'weight' in n], 'lr': 0},
        {'params': net.fuse.weight, 'lr': lr / 100, 'weight_decay': wd},
        {'params': net.fuse.bias, 'lr': 2 * lr / 100},
    ]
    if os.environ.get("OSVOS_FUSED_SGD", "1") != "0":
        from osvos_pytorch_amd.optim import FusedSGD
        return FusedSGD(groups, lr=lr, momentum=0.9)
    return torch.optim.SGD(groups, lr=lr, momentum=0.9)


def cpu_baseline(h, w, mode, n_ave, budget_s=20.0):
    """The reference loop restated on torch CPU (oracle/torch_ref.py: the same ATen/oneDNN kernels
    the reference executes), timed on this node's host cores on a bounded sample."""
    from oracle import synth, torch_ref
    # pick the thread count the host runs a mid-size conv fastest with (all 256 hardware threads of
    # the GPU node oversubscribe oneDNN badly); report the count actually used as `cores`
    ncpu = os.cpu_count() or 1
    probe_x, probe_w = torch.randn(1, 256, 120, 214), torch.randn(256, 256, 3, 3)
    best_t, best_n = 1e9, ncpu
    for nt in sorted({ncpu, max(1, ncpu // 2), max(1, ncpu // 4), 64, 32, 16, 8} & set(range(1, ncpu + 1))):
        torch.set_num_threads(nt)
        torch.nn.functional.conv2d(probe_x, probe_w, padding=1)
        t0 = time.perf_counter()
        for _ in range(3):
            torch.nn.functional.conv2d(probe_x, probe_w, padding=1)
        dt = time.perf_counter() - t0
        if dt < best_t:
            best_t, best_n = dt, nt
    torch.set_num_threads(best_n)
    wts = synth.make_weights(1)
    x = torch.from_numpy(synth.make_frame(1, h, w, 0))
    m = torch.from_numpy(synth.make_mask(1, h, w, 0))
    p = torch_ref.as_leaf_params(wts)
    opt = torch.optim.SGD(torch_ref.sgd_groups(p, mode=mode), lr=1e-8, momentum=0.9)
    times = []
    t_start = time.perf_counter()
    it = 0
    while True:
        t0 = time.perf_counter()
        loss, _ = torch_ref.train_loss(p, x.clone().requires_grad_(), m, mode=mode)
        _ = loss.item()
        (loss / n_ave).backward()
        it += 1
        if it % n_ave == 0:
            opt.step()
            opt.zero_grad()
        times.append(time.perf_counter() - t0)
        if (time.perf_counter() - t_start > budget_s and len(times) >= 3) or len(times) >= 12:
            break
    timed = times[1:] if len(times) > 1 else times     # first iteration = warm-up
    med = float(np.median(timed))
    return {"value": 1.0 / med, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d fwd+bwd iterations of the restated train_%s.py loop at %dx%d, batch 1, fp32, torch %s CPU "
                      "(oneDNN), median after 1 warm-up" % (len(timed), mode, w, h, torch.__version__)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", default="online", choices=["online", "parent", "infer"],
                    help="online/parent: restated training loops (fwd+loss+bwd+SGD); infer: forward only under no_grad "
                         "(BASELINE.json configs[4]: use --height 1080 --width 1920 --batch 4 --graph 1)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"],
                    help="bf16: the three conv passes on bf16 MFMA operands, trunk tensors stored as bf16 (fp32 accumulate); "
                         "the headline configs[1] is fp32")
    ap.add_argument("--graph", type=int, default=0, help="infer mode: replay the forward from a captured hipGraph")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=854)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--n-ave-grad", type=int, default=0, help="0 = reference value (5 online, 10 parent)")
    ap.add_argument("--item-sync", type=int, default=0, help="1 = loss.item() every iteration like the reference's logging")
    ap.add_argument("--force-dist", action="store_true", help="initialise the nccl (RCCL) process group and run the gradient "
                    "all-reduce even with one rank (single-GPU check of the multi-GPU path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        dist = None
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from layers.osvos_layers import class_balanced_cross_entropy_loss as cbce
    from osvos_pytorch_amd import _lib
    from osvos_pytorch_amd.parallel import GradientAllReducer

    n_ave = args.n_ave_grad or (5 if args.mode == "online" else 10)
    net, x, gt = synth_problem(args.batch, args.height, args.width, device, seed=rank)
    net.set_precision(args.precision)
    opt = make_optimizer(net, "online" if args.mode == "infer" else args.mode)
    reducer = GradientAllReducer(net, average=True, always=args.force_dist) if dist is not None else None
    running = torch.zeros((), device=device)
    state = {"ave": 0, "epoch": 0}

    def step():
        # body of train_online.py:116-149 (train_parent.py:132-172 for --mode parent)
        inputs = x.detach().requires_grad_()         # train_online.py:121: the input gradient is computed
        outputs = net.forward(inputs)
        if args.mode == "online":
            loss = cbce(outputs[-1], gt, size_average=False)
        else:
            losses = [cbce(o, gt, size_average=False) for o in outputs]
            loss = (1 - state["epoch"] / 240) * sum(losses[:-1]) + losses[-1]
        if args.item_sync:
            running.add_(loss.item())
        else:
            running.add_(loss.detach())
        loss /= n_ave
        loss.backward()
        state["ave"] += 1
        if state["ave"] % n_ave == 0:
            if reducer is not None:
                reducer.all_reduce()
            opt.step()
            if reducer is not None:
                reducer.zero_grads()
            else:
                opt.zero_grad()
            state["ave"] = 0

    if args.mode == "infer":
        keep = {}

        def infer_eager():
            with torch.no_grad():
                keep["outs"] = net.forward(x)      # train_online.py:172-181 (sigmoid/PNG writing is host I/O)
        step = infer_eager
        if args.graph:
            for _ in range(3):
                infer_eager()                      # packs weights, creates the aux stream/events, sets kernel attributes
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                infer_eager()
            step = graph.replay
    for _ in range(args.warmup):
        step()
    lib = _lib.lib()
    prof = (not args.no_prof) and not (args.mode == "infer" and args.graph)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    if prof:
        _lib.check(lib.osvos_prof_start(args.steps * 64 + 64), "prof_start")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ms = (C.c_double * 4)()
    fl = (C.c_double * 4)()
    cnt = (C.c_long * 4)()
    if prof:
        _lib.check(lib.osvos_prof_stop(ms, fl, cnt), "prof_stop")
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    frames = args.steps * args.batch * world
    value = frames / elapsed

    if rank == 0:
        gf_fwd = conv_gflop_forward(args.height, args.width) * args.batch
        passes = 1 if args.mode == "infer" else 3
        peak = FP32_MFMA_PEAK_TFLOPS if args.precision == "fp32" else BF16_MFMA_PEAK_TFLOPS
        kname = ("conv3x3_f32_kernel", "wgrad_f32_kernel") if args.precision == "fp32" else ("conv3x3_bf16_kernel", "wgrad_bf16_kernel")
        roof = None
        if args.mode == "infer" and args.graph:
            # one captured graph per step: the family is the whole forward (17 conv launches + glue)
            ach = gf_fwd / 1e3 / (elapsed / args.steps)
            act_gb = 0.904 * (args.height * args.width) / (480.0 * 854.0) * args.batch     # min conv tensor traffic, fp32
            roof = {"bound": "mfma", "kernel": "hipGraph replay of osvos_net_forward (%s x17 + pool/head glue)" % kname[0],
                    "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": None, "algorithmic_hbm_GBps": round(act_gb / (elapsed / args.steps), 1), "hbm_peak_GBps": 8000}
        if prof and cnt[0] + cnt[1] > 0:
            # dominant kernel family: the MFMA conv kernels.  Family 0 = conv3x3_f32_kernel forward
            # launches (one event pair per launch); family 1 = backward regions, i.e. the data-gradient
            # launch of a layer running concurrently with its weight-gradient launch (+ slab reduce) on
            # the second stream, timed fork -> join.  FLOPs are algorithmic (2*N*H*W*Cout*9*Cin each).
            conv_ms = ms[0] + ms[1]
            conv_fl = fl[0] + fl[1]
            ach = conv_fl / (conv_ms * 1e-3) / 1e12
            roof = {"bound": "mfma", "kernel": "%s fwd launches + (%s dgrad || %s) backward regions" % (kname[0], kname[0], kname[1]),
                    "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4),
                    "launches": int(cnt[0] + cnt[1]), "avg_launch_ms": round(conv_ms / (cnt[0] + cnt[1]), 4),
                    "algorithmic_gflop_per_launch": round(conv_fl / (cnt[0] + cnt[1]) / 1e9, 3),
                    "families": {"conv_fwd": {"ms_per_step": round(ms[0] / args.steps, 3), "tflops": round(fl[0] / (ms[0] * 1e-3) / 1e12, 2) if ms[0] else None},
                                 "conv_bwd_dgrad+wgrad": {"ms_per_step": round(ms[1] / args.steps, 3), "tflops": round(fl[1] / (ms[1] * 1e-3) / 1e12, 2) if ms[1] else None}},
                    "step_conv_fraction_of_mfma_roofline": round(passes * gf_fwd / 1e3 / (elapsed / args.steps) / peak, 4),
                    # HBM-side bytes per launch from rocprofv3 --pmc (profiles/r01_pmc_conv3_2_conv1_2.txt), conv3_2 forward
                    # launch: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE; algorithmic 55.0 MB
                    "traffic": {"conv3_2_fwd_launch_MB": 94.1, "algorithmic_MB": 55.0, "source": "profiles/r01_pmc_conv3_2_conv1_2.txt"}}
        base = None
        if not args.no_cpu_baseline and world == 1 and args.mode != "infer":
            try:
                base = cpu_baseline(args.height, args.width, args.mode, n_ave)
            except Exception as e:  # the GPU result must still be reported
                base = {"error": repr(e)}
        line = {
            "metric": ("frames/sec (fwd+bwd) OSVOS-VGG16 854x480 per GPU" if (args.height, args.width) == (480, 854) else
                       "frames/sec (fwd+bwd) OSVOS-VGG16 %dx%d" % (args.width, args.height)) if args.mode != "infer" else
                      "frames/sec (forward only) OSVOS-VGG16 %dx%d" % (args.width, args.height),
            "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "bf16 MFMA operands and bf16 trunk tensors (fwd+dgrad+wgrad), f32 accumulate; head/loss/skinny wgrads/parameters f32",
            "data": "synthetic",
            "config": {"workload": ("%dx%d batch=%d inference forward (train_online.py:172-181), no_grad, %s, fp32, frames resident in HBM"
                                    % (args.width, args.height, args.batch, "hipGraph replay" if args.graph else "eager launches"))
                       if args.mode == "infer" else
                       "%dx%d batch=%d %s fine-tune loop (train_%s.py), fused-head%s loss, nAveGrad=%d, SGD 8-group, "
                       "fp32, frame resident in HBM" % (args.width, args.height, args.batch, args.mode, args.mode,
                                                        "" if args.mode == "online" else "+4 side", n_ave),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                       "grad_allreduce": "per optimizer step (RCCL)" if dist is not None else "none",
                       "loss_item_sync_each_iter": bool(args.item_sync)},
            "roofline": roof, "cpu_baseline": base,
            "running_loss": float(running.item()) / max(1, args.steps + args.warmup),
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

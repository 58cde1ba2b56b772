#!/usr/bin/env python3
"""Where train_parent.py --device-augment spends its time per frame: the feeder alone (decode -> pinned -> H2D -> augmentation kernel), the
training micro-batch alone (frame resident), and both together -- per-frame milliseconds over N frames, after one warm pass."""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # like train_parent.py
import networks.vgg_osvos as vo  # noqa: E402
import train_parent as tp  # noqa: E402
from osvos_pytorch_amd.augment import DeviceAugment  # noqa: E402
from osvos_pytorch_amd.davis_io import DevicePrefetcher  # noqa: E402
from osvos_pytorch_amd.train_common import TrainLoop, make_sgd  # noqa: E402

n, h, w = 64, 480, 854
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda", 0)
frames = tp.synthetic_raw_frames(n, h, w)
aug = DeviceAugment(rots=(-30, 30), scales=(.75, 1.25))
if os.environ.get("PROBE_INIT", "calibrated") == "calibrated":      # He-init + calibrated heads (bench.py's weights): activations of realistic size
    import bench
    net, _, _ = bench.synth_problem(1, h, w, dev, seed=0)
else:                                                                  # the reference's N(0, 0.001) initialisation: activations ~1e-10 and below
    net = vo.OSVOS(pretrained=0).to(dev)
net.set_precision(prec)
loop = TrainLoop(net, make_sgd(net, "parent"), mode="parent", n_ave_grad=10)


def feeder():
    for _, img, lab in DevicePrefetcher(frames, list(range(n)), dev, depth=3):
        s = aug(img, lab)
        yield s['image'][None], s['gt'][None]


def timed(fn, label):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    print("%-62s %.3f ms per frame" % (label, (time.perf_counter() - t0) / n * 1e3))


def only_feed():
    for x, m in feeder():
        pass


keep = next(iter(feeder()))


def only_train():
    for _ in range(n):
        loop.micro_batch(keep[0].clone().requires_grad_(), keep[1], epoch=0)


def both():
    for x, m in feeder():
        loop.micro_batch(x.requires_grad_(), m, epoch=0)


print("weights:", os.environ.get("PROBE_INIT", "calibrated"))
timed(only_feed, "feeder alone (decode, pinned copy, H2D, augmentation kernel)")
timed(only_train, "training micro-batch alone (frame resident, %s)" % prec)
timed(both, "feeder + training (what train_parent.py --device-augment runs)")

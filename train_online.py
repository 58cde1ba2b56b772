"""Online fine-tuning on the first frame of a sequence, then inference on every frame.

Same entry point, knobs and file naming as the reference's train_online.py (SEQ_NAME env var,
parent checkpoint ``<save_dir>/parent_epoch-239.pth``, result PNGs under ``<save_dir>/Results/<seq>``),
running on the MI355X-native OSVOS path.  Differences by design:
  * the dataset / augmentation layer of the reference needs OpenCV (``cv2``), which this image
    lacks: ``--device-augment`` replaces it (Pillow decode -> pinned uint8 staging -> one HIP kernel
    for flip / scale+rotate / mean / CHW, osvos_pytorch_amd.augment; the first frame is decoded once
    and re-augmented on the GPU every iteration); ``--synthetic`` runs the identical loop on a seeded
    synthetic frame (benchmarking); with the reference's dataloaders package + cv2 installed next to
    this file the original transform chain is used
  * the loss is accumulated on the device and read back only when it is printed
  * launched under torchrun with N processes, rank r fine-tunes sequences r, r+N, ... of the
    comma-separated SEQ_NAME list (independent replicas: online training has no exchange step)
"""
from __future__ import division

import argparse
import os
import sys
import timeit

# ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The backward uses three streams (data gradients / weight
# gradients / slab reduces); once a RCCL communicator adds its own, two of ours share a hardware queue and serialise (bench.py
# --force-dist: 217 vs 224 frames/s; this script with OSVOS_DP_FORCE=1: 2.169 vs 2.143 s per 512-frame epoch).  Eight queues restore it;
# without a communicator 4 and 8 measure the same.  A WARNING that cost this script 39 % for most of round 4: a stream that carries only
# H2D copies must not get a hardware queue of its own next to these -- with the input pipeline's former copy stream on a 5th queue every
# step stretched from 4.3 to 6.2 ms (profiles/r04_scripts_e2e.txt); the pipeline now copies on the consumer's stream (davis_io.py).
# Must be set before the HIP runtime initialises, i.e. before the first CUDA call of the process.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

import networks.vgg_osvos as vo
from layers.osvos_layers import sigmoid_np  # noqa: F401  (kept importable like the reference)
from osvos_pytorch_amd.results import davis_statistics, jaccard, save_masks
from mypath import Path
from osvos_pytorch_amd.parallel import shard_indices
from osvos_pytorch_amd.train_common import TrainLoop, init_distributed, make_sgd


def synthetic_loader(h, w, seed):
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(1, 3, h, w, generator=g) * 40.0
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
    gt = ((((yy - 0.5 * h) / (0.25 * h)) ** 2 + ((xx - 0.5 * w) / (0.25 * w)) ** 2) <= 1).float()[None, None]
    return [{'image': img, 'gt': gt, 'fname': ['00000']}]


def davis_loaders(db_root_dir, seq_name):
    try:
        from torchvision import transforms
        from torch.utils.data import DataLoader
        from dataloaders import davis_2016 as db
        from dataloaders import custom_transforms as tr
    except ImportError as e:
        raise SystemExit("DAVIS loading needs the reference's dataloaders package + cv2 + torchvision (%s); "
                         "use --synthetic to run without data" % e)
    composed = transforms.Compose([tr.RandomHorizontalFlip(), tr.ScaleNRotate(rots=(-30, 30), scales=(.75, 1.25)), tr.ToTensor()])
    db_train = db.DAVIS2016(train=True, db_root_dir=db_root_dir, transform=composed, seq_name=seq_name)
    db_test = db.DAVIS2016(train=False, db_root_dir=db_root_dir, transform=tr.ToTensor(), seq_name=seq_name)
    return DataLoader(db_train, batch_size=1, shuffle=True, num_workers=1), DataLoader(db_test, batch_size=1, shuffle=False, num_workers=1)


class DeviceTrainFrame(object):
    """train_online.py:92-97 on the device: the sequence's first frame + annotation live on the GPU as uint8; every pass through
    the 'loader' draws the reference's random flip / rotation / scale (same order, Python's random module) and runs one HIP kernel."""

    def __init__(self, img_u8, lab_u8, device):
        from osvos_pytorch_amd.augment import DeviceAugment
        self.img, self.lab = torch.from_numpy(img_u8).to(device), torch.from_numpy(lab_u8).to(device)
        self.aug = DeviceAugment(rots=(-30, 30), scales=(.75, 1.25))

    def __len__(self):
        return 1

    def __iter__(self):
        s = self.aug(self.img, self.lab)
        yield {'image': s['image'][None], 'gt': s['gt'][None]}


class DeviceTestFrames(object):
    """train_online.py:98-100 on the device: every frame of the sequence, decoded on the host a few frames ahead, mean-subtracted and
    laid out CHW by the augmentation kernel with the identity transform (the reference's test transform is ToTensor only)."""

    def __init__(self, frames, device, depth):
        self.frames, self.device, self.depth = frames, device, depth

    def __len__(self):
        return len(self.frames)

    def __iter__(self):
        from osvos_pytorch_amd.augment import augment_frame
        from osvos_pytorch_amd.davis_io import DevicePrefetcher
        for idx, img, lab in DevicePrefetcher(self.frames, range(len(self.frames)), self.device, depth=self.depth):
            image, gt = augment_frame(img, lab, flip=False, rot=None)
            out = {'image': image[None], 'fname': [self.frames.fname(idx)]}
            if lab is not None:
                out['gt'] = gt[None]
            yield out


def device_loaders(args, seq_name, device, seed):
    import random
    from osvos_pytorch_amd.davis_io import ArrayFrames, DavisFrames
    random.seed(seed)
    if args.synthetic:
        s = synthetic_loader(args.height, args.width, seed)[0]
        img = (s['image'][0].permute(1, 2, 0) + 116.0).clamp(0, 255).to(torch.uint8).numpy()
        lab = (s['gt'][0, 0] * 255).to(torch.uint8).numpy()
        train, test = ArrayFrames([(img, lab)]), ArrayFrames([(img, lab)])
    else:
        train, test = DavisFrames(True, Path.db_root_dir(), seq_name=seq_name), DavisFrames(False, Path.db_root_dir(), seq_name=seq_name)
    img, lab = train[0]
    return DeviceTrainFrame(img, lab, device), DeviceTestFrames(test, device, args.prefetch)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--synthetic', action='store_true', help='seeded synthetic 854x480 frame instead of DAVIS')
    ap.add_argument('--epochs', type=int, default=0, help='0 = reference value 2000 * nAveGrad')
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=854)
    ap.add_argument('--device-augment', action='store_true',
                    help='input pipeline on the GPU: Pillow decode -> pinned uint8 -> osvos_augment_frame (flip, scale+rotate, mean, CHW); '
                         'the first frame is decoded ONCE and re-augmented on the device every iteration')
    ap.add_argument('--prefetch', type=int, default=3, help='--device-augment: test frames decoded / copied ahead of the forward')
    ap.add_argument('--precision', default=os.environ.get('OSVOS_PRECISION', 'fp32x3'), choices=['fp32', 'fp32x3', 'fp32x3b2', 'fp32x3h2', 'fp32h2', 'fp32x2', 'bf16'])
    ap.add_argument('--test-precision', default=os.environ.get('OSVOS_TEST_PRECISION', ''), choices=['', 'fp32', 'fp32x3', 'fp32h2', 'bf16'],
                    help="precision of the TEST forwards (train_online.py:172-189 of the reference; default: the training precision).  'fp32h2' -- the f32x3 "
                         "convolutions on two FP16 pieces under block exponents -- runs a forward 1.5x as fast as 'fp32x3' with logits closer to float64 "
                         "(DESIGN.md 3.1a); the masks are the same up to pixels within 1e-5 std of the threshold")
    ap.add_argument('--window-fused', action='store_true',
                    help='run the nAveGrad micro-batches of every optimizer step as ONE batch with per-image class counts (TrainLoop.window_batch): '
                         'the same gradient up to fp32 summation order, one set of kernel launches per optimizer step instead of nAveGrad')
    args = ap.parse_args()

    rank, world, device = init_distributed(collectives=False)      # sequences are sharded over the ranks: nothing is exchanged
    seqs = os.environ.get('SEQ_NAME', 'blackswan').split(',')
    save_dir = Path.save_root_dir()
    os.makedirs(save_dir, exist_ok=True)
    nAveGrad = 5
    nEpochs = args.epochs or 2000 * nAveGrad
    snapshot = nEpochs
    parentEpoch = 240
    seed = 0

    for si in shard_indices(len(seqs), rank, world):
        seq_name = seqs[si]
        net = vo.OSVOS(pretrained=0)
        parent = os.path.join(save_dir, 'parent_epoch-' + str(parentEpoch - 1) + '.pth')
        if os.path.exists(parent):
            net.load_state_dict(torch.load(parent, map_location=lambda storage, loc: storage))
        elif not args.synthetic:
            raise SystemExit('parent model %s not found' % parent)
        net.to(device)
        net.set_precision(args.precision)
        optimizer = make_sgd(net, 'online')
        if args.device_augment:
            trainloader, testloader = device_loaders(args, seq_name, device, seed + si)
        elif args.synthetic:
            trainloader = testloader = synthetic_loader(args.height, args.width, seed + si)
        else:
            trainloader, testloader = davis_loaders(Path.db_root_dir(), seq_name)
        loop = TrainLoop(net, optimizer, mode='online', n_ave_grad=nAveGrad)
        num_img_tr = len(trainloader)
        print('Start of Online Training, sequence: ' + seq_name)
        start_time = timeit.default_timer()
        window = []                      # --window-fused: the micro-batches of the open optimizer-step window
        for epoch in range(0, nEpochs):
            np.random.seed(seed + epoch)
            for ii, sample in enumerate(trainloader):
                inputs, gts = sample['image'], sample['gt']
                inputs, gts = inputs.to(device), gts.to(device)
                if args.window_fused:
                    window.append((inputs, gts))
                    if len(window) == nAveGrad:
                        loop.window_batch(torch.cat([w[0] for w in window]).requires_grad_(), torch.cat([w[1] for w in window]))
                        window = []
                    continue
                inputs = inputs.detach().requires_grad_()      # (a fresh leaf: a frame the loader hands out again must not accumulate a .grad)
                loop.micro_batch(inputs, gts)
            if epoch % max(1, nEpochs // 20) == max(1, nEpochs // 20) - 1:
                running = loop.pop_running()[0] / (num_img_tr * max(1, nEpochs // 20))
                print('[Epoch: %d, numImages: %5d]' % (epoch + 1, num_img_tr))
                print('Loss: %f' % running)
            if (epoch % snapshot) == snapshot - 1 and epoch != 0:
                torch.save(net.state_dict(), os.path.join(save_dir, seq_name + '_epoch-' + str(epoch) + '.pth'))
        if device.type == 'cuda':
            torch.cuda.synchronize()
        print('Online training time: ' + str(timeit.default_timer() - start_time))

        save_dir_res = os.path.join(save_dir, 'Results', seq_name)
        os.makedirs(save_dir_res, exist_ok=True)
        print('Testing Network')
        js = []
        if args.test_precision:
            net.set_precision(args.test_precision)      # (re-packs the weights once: the FP16-pair packs are another format)
        with torch.no_grad():
            for sample in testloader:
                img, fname = sample['image'], sample['fname']
                outputs = net.forward(img.to(device))
                # sigmoid + scipy<=1.1 imsave byte scaling on the device, PNG written by osvos_pytorch_amd.results (reference :181-187)
                save_masks(outputs[-1], [os.path.join(save_dir_res, os.path.basename(fname[jj]) + '.png') for jj in range(int(img.size()[0]))])
                if 'gt' in sample:
                    js.extend(jaccard(outputs[-1], sample['gt'].to(device)))
        if js:
            st = davis_statistics(js)
            print('J (region similarity) on %s: mean %.4f recall %.4f decay %.4f over %d frames' % (seq_name, st['mean'], st['recall'], st['decay'], len(js)))


if __name__ == '__main__':
    sys.exit(main())

"""Parent-network training (deep supervision over the 5 heads), same knobs and checkpoint naming as
the reference's train_parent.py (240 epochs, nAveGrad 10, snapshot every 40 epochs to
``<save_dir>/parent_epoch-<e>.pth``), on the MI355X-native OSVOS path.

Data parallel: launch with ``python -m torch.distributed.run --nproc-per-node N train_parent.py``;
rank r takes micro-batches r, r+N, ... of every optimizer step, accumulates locally and the flat
gradient buffer is all-reduced (RCCL over xGMI) once per step.  With the reference's batch-1,
per-frame class weights this reproduces the single-process gradient exactly (up to summation
order) whenever N divides nAveGrad (N in {1, 2, 5, 10}); for N in {4, 8} use --n-ave-grad 8 / 16.
"""
from __future__ import division

import argparse
import os
import sys
import timeit

# ROCm maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The backward uses three streams (data gradients /
# weight gradients / slab reduces); once torch.distributed's RCCL communicator adds its own streams two of ours end up on the same
# hardware queue and serialise -- measured -7 % (128 -> 119 frames/s) from init_process_group alone.  Eight queues restore it.
# Must be set before the HIP runtime initialises, i.e. before the first CUDA call of the process.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch

import networks.vgg_osvos as vo
from mypath import Path
from osvos_pytorch_amd.train_common import TrainLoop, init_distributed, make_reducer, make_sgd


def synthetic_dataset(n, h, w):
    out = []
    for i in range(n):
        g = torch.Generator().manual_seed(100 + i)
        img = torch.randn(1, 3, h, w, generator=g) * 40.0
        yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing='ij')
        cy, cx = 0.3 + 0.4 * torch.rand(1, generator=g).item(), 0.3 + 0.4 * torch.rand(1, generator=g).item()
        gt = ((((yy - cy * h) / (0.2 * h)) ** 2 + ((xx - cx * w) / (0.2 * w)) ** 2) <= 1).float()[None, None]
        out.append({'image': img, 'gt': gt})
    return out


def davis_loaders(db_root_dir):
    try:
        from torchvision import transforms
        from torch.utils.data import DataLoader
        from dataloaders import davis_2016 as db
        from dataloaders import custom_transforms as tr
    except ImportError as e:
        raise SystemExit("DAVIS loading needs the reference's dataloaders package + cv2 + torchvision (%s); "
                         "use --synthetic to run without data" % e)
    composed = transforms.Compose([tr.RandomHorizontalFlip(), tr.ScaleNRotate(rots=(-30, 30), scales=(.75, 1.25)), tr.ToTensor()])
    db_train = db.DAVIS2016(train=True, inputRes=None, db_root_dir=db_root_dir, transform=composed)
    db_test = db.DAVIS2016(train=False, db_root_dir=db_root_dir, transform=tr.ToTensor())
    return DataLoader(db_train, batch_size=1, shuffle=True, num_workers=2), DataLoader(db_test, batch_size=1, shuffle=False, num_workers=2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--synthetic', type=int, default=0, help='N > 0: train on N seeded synthetic frames instead of DAVIS')
    ap.add_argument('--epochs', type=int, default=240)
    ap.add_argument('--n-ave-grad', type=int, default=10)
    ap.add_argument('--resume-epoch', type=int, default=0)
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=854)
    args = ap.parse_args()

    rank, world, device = init_distributed()
    nEpochs, nAveGrad, resume_epoch = args.epochs, args.n_ave_grad, args.resume_epoch
    snapshot, nTestInterval = 40, 5
    save_dir = Path.save_root_dir()
    os.makedirs(save_dir, exist_ok=True)
    modelName = 'parent'

    if resume_epoch == 0:
        have_caffe = os.path.exists(os.path.join(Path.models_dir(), 'vgg_caffe.mat'))
        have_pt = os.path.exists(os.path.join(Path.models_dir(), 'vgg_pytorch.pth'))
        net = vo.OSVOS(pretrained=2 if have_caffe else (1 if have_pt else 0))
    else:
        net = vo.OSVOS(pretrained=0)
        ckpt = os.path.join(save_dir, modelName + '_epoch-' + str(resume_epoch - 1) + '.pth')
        print("Updating weights from: {}".format(ckpt))
        net.load_state_dict(torch.load(ckpt, map_location=lambda storage, loc: storage))
    net.to(device)
    optimizer = make_sgd(net, 'parent')
    reducer = make_reducer(net, world, average=False)
    if reducer is not None:
        reducer.broadcast_parameters(0)
    if args.synthetic:
        trainset, testset = synthetic_dataset(args.synthetic, args.height, args.width), synthetic_dataset(2, args.height, args.width)
    else:
        trainset, testset = davis_loaders(Path.db_root_dir())
    # every rank walks the same order and keeps the micro-batches r, r+W, ... of each group of nAveGrad
    local_ave = max(1, nAveGrad // world)
    loop = TrainLoop(net, optimizer, mode='parent', n_ave_grad=nAveGrad, n_epochs=nEpochs, reducer=reducer)
    loop.n_ave_grad = nAveGrad          # loss divisor stays the global nAveGrad (sum over all ranks)
    print("Training Network")
    for epoch in range(resume_epoch, nEpochs):
        start_time = timeit.default_timer()
        count = 0
        for ii, sample in enumerate(trainset):
            if ii % world != rank:
                continue
            inputs, gts = sample['image'], sample['gt']
            inputs.requires_grad_()
            inputs, gts = inputs.to(device), gts.to(device)
            outputs = net.forward(inputs)
            from layers.osvos_layers import class_balanced_cross_entropy_loss as cbce
            losses = [cbce(o, gts, size_average=False) for o in outputs]
            for r, l in zip(loop.running, losses):
                r += l.detach()
            loss = (1 - epoch / nEpochs) * sum(losses[:-1]) + losses[-1]
            loss /= nAveGrad
            loss.backward()
            count += 1
            if count % local_ave == 0:
                if reducer is not None:
                    reducer.all_reduce()
                optimizer.step()
                optimizer.zero_grad()
        running = [v / max(1, count) for v in loop.pop_running()]
        if rank == 0:
            print('[Epoch: %d, numImages: %5d]' % (epoch, count * world))
            for l, v in enumerate(running):
                print('Loss %d: %f' % (l, v))
            print("Execution time: " + str(timeit.default_timer() - start_time))
        if (epoch % snapshot) == snapshot - 1 and epoch != 0 and rank == 0:
            torch.save(net.state_dict(), os.path.join(save_dir, modelName + '_epoch-' + str(epoch) + '.pth'))
        if epoch % nTestInterval == (nTestInterval - 1):
            with torch.no_grad():
                tot = [0.0] * 5
                for sample in testset:
                    outputs = net.forward(sample['image'].to(device))
                    from layers.osvos_layers import class_balanced_cross_entropy_loss as cbce
                    for i, o in enumerate(outputs):
                        tot[i] += cbce(o, sample['gt'].to(device), size_average=False).item()
                if rank == 0:
                    for l, v in enumerate(tot):
                        print('***Testing *** Loss %d: %f' % (l, v / max(1, len(testset))))


if __name__ == '__main__':
    sys.exit(main())

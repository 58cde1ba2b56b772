"""Parent-network training (deep supervision over the 5 heads), same knobs and checkpoint naming as
the reference's train_parent.py (240 epochs, nAveGrad 10, snapshot every 40 epochs to
``<save_dir>/parent_epoch-<e>.pth``), on the MI355X-native OSVOS path.

Data parallel: launch with ``python -m torch.distributed.run --nproc-per-node N train_parent.py``.
Every rank derives the same per-epoch permutation of the frames (``--seed``) and runs the iterations
g = rank (mod N) of that global stream -- indices are sharded BEFORE anything is decoded; the stream is
cut into optimizer steps of nAveGrad consecutive iterations (across epoch boundaries, like the
reference's persistent ``aveGrad`` counter), each rank accumulates its nAveGrad / N micro-batches
locally and the flat gradient buffer is all-reduced (RCCL over xGMI) once per step.  With the
reference's batch-1, per-frame class weights this reproduces the single-process gradient of the same
stream exactly (up to summation order).  N must divide nAveGrad (N in {1, 2, 5, 10} for the
reference's 10); for N in {4, 8} pass --n-ave-grad 8 / 16 -- anything else is refused, not approximated.

``--device-augment`` replaces the reference's cv2 transform chain (train_parent.py:106-110) by Pillow
decode -> pinned uint8 staging -> one HIP kernel (osvos_pytorch_amd.augment), prefetched on a copy stream.
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
from osvos_pytorch_amd.train_common import StepSchedule, TrainLoop, check_world_divides, epoch_plan, init_distributed, make_reducer, make_sgd


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


def synthetic_raw_frames(n, h, w):
    """uint8 BGR frames + 0/255 labels for --device-augment --synthetic (what cv2.imread would hand the reference)."""
    from osvos_pytorch_amd.davis_io import ArrayFrames
    frames = []
    for s in synthetic_dataset(n, h, w):
        img = (s['image'][0].permute(1, 2, 0) + 116.0).clamp(0, 255).to(torch.uint8).numpy()
        frames.append((img, (s['gt'][0, 0] * 255).to(torch.uint8).numpy()))
    return ArrayFrames(frames)


def davis_datasets(db_root_dir):
    """The reference's datasets (train_parent.py:106-113), indexable: the data-parallel plan shards INDICES before anything is
    decoded, so no rank ever loads a frame it does not train on."""
    try:
        from torchvision import transforms
        from dataloaders import davis_2016 as db
        from dataloaders import custom_transforms as tr
    except ImportError as e:
        raise SystemExit("DAVIS loading through the reference's transforms needs its dataloaders package + cv2 + torchvision (%s); "
                         "use --device-augment (Pillow decode + HIP augmentation) or --synthetic N" % e)
    composed = transforms.Compose([tr.RandomHorizontalFlip(), tr.ScaleNRotate(rots=(-30, 30), scales=(.75, 1.25)), tr.ToTensor()])
    db_train = db.DAVIS2016(train=True, inputRes=None, db_root_dir=db_root_dir, transform=composed)
    db_test = db.DAVIS2016(train=False, db_root_dir=db_root_dir, transform=tr.ToTensor())
    return db_train, db_test


def epoch_samples(args, trainset, plan, device, augment):
    """Yield {'image': [1,3,H,W], 'gt': [1,1,H,W]} for this rank's frames of one epoch, in plan order."""
    indices = [idx for idx, _ in plan]
    if augment is not None:
        # decode on the host -> pinned uint8 -> GPU (copy stream, a few frames ahead) -> one HIP kernel: mean / flip / warp / CHW
        from osvos_pytorch_amd.davis_io import DevicePrefetcher
        for _, img, lab in DevicePrefetcher(trainset, indices, device, depth=args.prefetch):
            s = augment(img, lab)
            yield {'image': s['image'][None], 'gt': s['gt'][None]}
    elif isinstance(trainset, list):
        for idx in indices:
            yield trainset[idx]
    else:
        from torch.utils.data import DataLoader
        for s in DataLoader(trainset, batch_size=1, sampler=indices, num_workers=2):
            yield s


def main(argv=None, build_net=None, loss_fn=None):
    """``argv`` / ``build_net`` / ``loss_fn``: the CPU tests of the data-parallel bookkeeping run THIS function on two gloo ranks with
    the network's forward and the loss swapped for the oracle's (tests/test_train_plan_cpu.py); the script passes none of them."""
    ap = argparse.ArgumentParser()
    ap.add_argument('--synthetic', type=int, default=0, help='N > 0: train on N seeded synthetic frames instead of DAVIS')
    ap.add_argument('--epochs', type=int, default=240)
    ap.add_argument('--n-ave-grad', type=int, default=10)
    ap.add_argument('--resume-epoch', type=int, default=0)
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=854)
    ap.add_argument('--seed', type=int, default=0, help='seed of the per-epoch frame permutation (identical on every rank)')
    ap.add_argument('--device-augment', action='store_true',
                    help='input pipeline on the GPU: Pillow decode -> pinned uint8 -> osvos_augment_frame (flip, scale+rotate, mean, CHW)')
    ap.add_argument('--prefetch', type=int, default=3, help='--device-augment: frames decoded / copied ahead of the training step')
    ap.add_argument('--precision', default=os.environ.get('OSVOS_PRECISION', 'fp32x3'), choices=['fp32', 'fp32x3', 'bf16'])
    ap.add_argument('--lr', type=float, default=1e-8, help='base learning rate of the SGD groups (train_parent.py:83)')
    ap.add_argument('--snapshot', type=int, default=40, help='store a model every this many epochs (train_parent.py:38)')
    ap.add_argument('--save-optimizer', action='store_true',
                    help='next to every parent_epoch-<e>.pth also write parent_epoch-<e>.optim.pth (SGD momentum buffers, the open accumulation '
                         'window: counters + gradients, pending epoch statistics, augmentation RNG; one .optim.rank<r>.pth per further rank) and, '
                         'with --resume-epoch, continue from it BIT FOR BIT.  The reference saves the network only and restarts momentum and '
                         'the window on resume (train_parent.py:59-65,175-176): that remains the behaviour without this flag.')
    args = ap.parse_args(argv)

    rank, world, device = init_distributed()
    nEpochs, nAveGrad, resume_epoch = args.epochs, args.n_ave_grad, args.resume_epoch
    local_ave = check_world_divides(nAveGrad, world)      # raises when the world size does not divide nAveGrad
    snapshot, nTestInterval = args.snapshot, 5
    save_dir = Path.save_root_dir()
    os.makedirs(save_dir, exist_ok=True)
    modelName = 'parent'

    if build_net is not None:
        net = build_net()
        if resume_epoch > 0:
            net.load_state_dict(torch.load(os.path.join(save_dir, modelName + '_epoch-' + str(resume_epoch - 1) + '.pth'), map_location='cpu'))
    elif resume_epoch == 0:
        have_caffe = os.path.exists(os.path.join(Path.models_dir(), 'vgg_caffe.mat'))
        have_pt = os.path.exists(os.path.join(Path.models_dir(), 'vgg_pytorch.pth'))
        net = vo.OSVOS(pretrained=2 if have_caffe else (1 if have_pt else 0))
    else:
        net = vo.OSVOS(pretrained=0)
        ckpt = os.path.join(save_dir, modelName + '_epoch-' + str(resume_epoch - 1) + '.pth')
        print("Updating weights from: {}".format(ckpt))
        net.load_state_dict(torch.load(ckpt, map_location=lambda storage, loc: storage))
    net.to(device)
    if build_net is None:
        net.set_precision(args.precision)
    optimizer = make_sgd(net, 'parent', lr=args.lr, fused=None if device.type == 'cuda' else False)
    reducer = make_reducer(net, world, average=False)
    if reducer is not None:
        reducer.broadcast_parameters(0)
        if hasattr(net, 'invalidate_packed_weights'):
            net.invalidate_packed_weights()  # broadcast writes through .data: the packed-weight cache cannot see it
    augment = None
    if args.device_augment:
        import random
        from osvos_pytorch_amd.augment import DeviceAugment
        from osvos_pytorch_amd.davis_io import DavisFrames
        random.seed(args.seed * 7919 + rank)                 # augmentation draws: independent per rank, reproducible
        augment = DeviceAugment(rots=(-30, 30), scales=(.75, 1.25))
        trainset = synthetic_raw_frames(args.synthetic, args.height, args.width) if args.synthetic else DavisFrames(True, Path.db_root_dir())
        testset = synthetic_dataset(2, args.height, args.width) if args.synthetic else None
    elif args.synthetic:
        trainset, testset = synthetic_dataset(args.synthetic, args.height, args.width), synthetic_dataset(2, args.height, args.width)
    else:
        trainset, testset = davis_datasets(Path.db_root_dir())
    # loss divisor = the global nAveGrad (sum over all ranks); this rank contributes nAveGrad / world micro-batches per step.
    # Optimizer steps straddle epoch boundaries (2079 frames, nAveGrad 10), so "the end of an epoch" is NOT a point at which the ranks
    # stand at the same place of their collective sequence: the schedule says after which gradient all-reduce an epoch's statistics
    # may be exchanged, and how many complete step windows the run has (no rank steps on the trailing partial one).
    def optim_path(e, r):
        return os.path.join(save_dir, modelName + '_epoch-' + str(e) + ('.optim.pth' if r == 0 else '.optim.rank%d.pth' % r))

    carry, resume_state = 0, None
    if args.save_optimizer and resume_epoch > 0 and os.path.exists(optim_path(resume_epoch - 1, 0)):
        # exact resume: optimizer state (identical on every rank: rank 0's file) + this rank's share of the open window
        common = torch.load(optim_path(resume_epoch - 1, 0), map_location='cpu', weights_only=False)
        resume_state = common if rank == 0 else torch.load(optim_path(resume_epoch - 1, rank), map_location='cpu', weights_only=False)
        if common['world'] != world:
            raise SystemExit("exact resume: %s was written by %d ranks, this run has %d (the open accumulation window is sharded by rank)"
                             % (optim_path(resume_epoch - 1, 0), common['world'], world))
        optimizer.load_state_dict(common['optimizer'])
        carry = int(common['carry'])
        print("Exact resume from %s: %d iteration(s) of the open window restored" % (optim_path(resume_epoch - 1, 0), carry))
    sched = StepSchedule(len(trainset), nAveGrad, resume_epoch, nEpochs, carry=carry)
    loop = TrainLoop(net, optimizer, mode='parent', n_ave_grad=nAveGrad, n_epochs=nEpochs, reducer=reducer, local_ave=local_ave,
                     loss_fn=loss_fn, max_steps=sched.total_steps)
    pending, started = [], {}          # epochs whose frames this rank has finished but whose statistics are not exchanged yet
    if resume_state is not None:
        loop.load_state_dict(resume_state['loop'])
        pending = sorted(int(e) for e in resume_state['loop']['counts'])      # epochs finished before the snapshot, statistics still to exchange
        for e in pending:
            started[e] = timeit.default_timer()
        if resume_state.get('py_random') is not None:
            import random
            random.setstate(resume_state['py_random'])

    def close_epochs(epochs):
        """Exchange + print the statistics of `epochs`: called by every rank right after the same gradient collective (or after the
        last epoch), so the small all-reduce below can never pair with another rank's gradient all-reduce."""
        for e in epochs:
            pending.remove(e)
            running, count = loop.pop_running(e), loop.pop_count(e)
            if reducer is not None and reducer.comm is not None:      # OSVOS_DP_BACKEND=abi: the C ABI's float64 sum
                t = torch.tensor(running + [float(count)], device=device, dtype=torch.float64)
                reducer.comm.all_reduce(t)
            elif reducer is not None:
                import torch.distributed as dist
                t = torch.tensor(running + [float(count)], device=device, dtype=torch.float64)
                dist.all_reduce(t)
            if reducer is not None:
                running, count = t[:-1].tolist(), int(round(float(t[-1].item())))
            if rank == 0:
                print('[Epoch: %d, numImages: %5d]' % (e, count))
                for l, v in enumerate(running):
                    print('Loss %d: %f' % (l, v / max(1, count)))
                print("Execution time: " + str(timeit.default_timer() - started[e]))

    print("Training Network")
    for epoch in range(resume_epoch, nEpochs):
        started[epoch] = timeit.default_timer()
        pending.append(epoch)          # (before its first frame: the window that closes the PREVIOUS epoch may end inside this one)
        # one permutation per epoch, the same on every rank; rank r runs the iterations g = r (mod world) of the global stream
        plan = epoch_plan(len(trainset), epoch, nAveGrad, rank, world, seed=args.seed)
        for sample in epoch_samples(args, trainset, plan, device, augment):
            inputs, gts = sample['image'], sample['gt']
            inputs.requires_grad_()                         # train_parent.py:136: the input gradient is computed
            inputs, gts = inputs.to(device), gts.to(device)
            _, stepped = loop.micro_batch(inputs, gts, epoch=epoch)      # forward, 5 losses, /= nAveGrad, backward, step every local_ave
            if stepped:
                close_epochs(sched.closed_by(loop.steps, pending))      # right after the SAME gradient collective on every rank
        if (epoch % snapshot) == snapshot - 1 and epoch != 0:
            if rank == 0:
                torch.save(net.state_dict(), os.path.join(save_dir, modelName + '_epoch-' + str(epoch) + '.pth'))
            if args.save_optimizer:
                import random
                st = {'loop': loop.state_dict(), 'world': world, 'py_random': random.getstate() if augment is not None else None,
                      'carry': (sched.carry + (epoch + 1 - resume_epoch) * len(trainset)) % nAveGrad}
                if rank == 0:
                    st['optimizer'] = optimizer.state_dict()
                torch.save(st, optim_path(epoch, rank))
        if testset is not None and epoch % nTestInterval == (nTestInterval - 1):
            from layers.osvos_layers import class_balanced_cross_entropy_loss as cbce
            with torch.no_grad():
                tot = [0.0] * 5
                for sample in testset:
                    outputs = net.forward(sample['image'].to(device))
                    for i, o in enumerate(outputs):
                        tot[i] += cbce(o, sample['gt'].to(device), size_average=False).item()
                if rank == 0:
                    for l, v in enumerate(tot):
                        print('***Testing *** Loss %d: %f' % (l, v / max(1, len(testset))))
    close_epochs(list(pending))        # epochs that end in the trailing partial window: every rank has left the loop, same order everywhere
    if rank == 0:
        print("optimizer steps taken: %d" % loop.steps)
    return net, loop


if __name__ == '__main__':
    main()
    sys.exit(0)

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
decode thread -> pinned uint8 ring -> H2D on the training stream -> one HIP kernel (osvos_pytorch_amd.augment), a few frames ahead.
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
        # decode on a host thread -> pinned uint8 ring (a few frames ahead) -> H2D on this stream -> one HIP kernel: mean / flip / warp / CHW
        from osvos_pytorch_amd.davis_io import DevicePrefetcher
        import random
        for (_, img, lab), (_, g) in zip(DevicePrefetcher(trainset, indices, device, depth=args.prefetch), plan):
            # the augmentation draws of a frame are a function of (seed, global iteration): the same whatever the number of ranks, and a
            # resumed run draws what the uninterrupted one would have
            random.seed(args.seed * 1000003 + g)
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
    ap.add_argument('--init-seed', type=int, default=None,
                    help='torch.manual_seed before the network is constructed (the reference leaves its N(0, 0.001) initialisation unseeded: default)')
    ap.add_argument('--device-augment', action='store_true',
                    help='input pipeline on the GPU: Pillow decode -> pinned uint8 -> osvos_augment_frame (flip, scale+rotate, mean, CHW)')
    ap.add_argument('--prefetch', type=int, default=3, help='--device-augment: frames decoded / copied ahead of the training step')
    ap.add_argument('--precision', default=os.environ.get('OSVOS_PRECISION', 'fp32x3'), choices=['fp32', 'fp32x3', 'fp32x3b2', 'fp32x3h2', 'fp32h2', 'fp32x2', 'bf16'])
    ap.add_argument('--lr', type=float, default=1e-8, help='base learning rate of the SGD groups (train_parent.py:83)')
    ap.add_argument('--snapshot', type=int, default=40, help='store a model every this many epochs (train_parent.py:38)')
    ap.add_argument('--test-interval', type=int, default=5, help='run the validation pass every this many epochs (train_parent.py:39)')
    ap.add_argument('--save-optimizer', action='store_true',
                    help='for every snapshot epoch also write parent_epoch-<e>.optim.pth: network, SGD momentum buffers and the position in the '
                         'global iteration stream, taken right after the optimizer step that closes the epoch (the one point at which all ranks '
                         'hold the same weights and no gradient is half accumulated); with --resume-epoch e+1 the run continues from it BIT FOR '
                         'BIT, on any number of ranks that divides nAveGrad.  The reference saves the network only and restarts momentum and '
                         'the accumulation window on resume (train_parent.py:59-65,175-176): that remains what a run WRITES without this flag.  '
                         'A bundle that exists for --resume-epoch is always used (see --no-resume-optimizer).')
    ap.add_argument('--no-resume-optimizer', action='store_true',
                    help="resume like the reference (network only: momentum and the open accumulation window restart) even when "
                         "parent_epoch-<e>.optim.pth exists")
    args = ap.parse_args(argv)

    rank, world, device = init_distributed()
    nEpochs, nAveGrad, resume_epoch = args.epochs, args.n_ave_grad, args.resume_epoch
    local_ave = check_world_divides(nAveGrad, world)      # raises when the world size does not divide nAveGrad
    snapshot, nTestInterval = args.snapshot, args.test_interval
    save_dir = Path.save_root_dir()
    os.makedirs(save_dir, exist_ok=True)
    modelName = 'parent'

    if args.init_seed is not None:
        torch.manual_seed(args.init_seed)
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
    def optim_path(e):
        return os.path.join(save_dir, modelName + '_epoch-' + str(e) + '.optim.pth')
    bundles_written = set()      # epochs whose bundle THIS run wrote (rank 0)

    # exact resume: the bundle holds network + optimizer + the global iteration the run continues at
    start_iteration, bundle = None, None
    # The bundle is used whenever it exists (a resume that silently dropped the momentum because a flag was not repeated would be a trap;
    # --no-resume-optimizer asks for the reference's network-only resume), and the ranks must AGREE: only rank 0 writes it, so without a shared
    # filesystem some ranks would take the exact resume and others the reference's -- different start iterations, diverging collective
    # sequences.  One small all-reduce of "I see it" settles it before anything depends on it.
    have = resume_epoch > 0 and not args.no_resume_optimizer and os.path.exists(optim_path(resume_epoch - 1))
    if reducer is not None:
        seen = torch.tensor([1.0 if have else 0.0], device=device, dtype=torch.float64)
        if reducer.comm is not None:
            reducer.comm.all_reduce(seen)
        else:
            import torch.distributed as dist
            dist.all_reduce(seen)
        if 0 < int(round(seen.item())) < world:
            raise SystemExit("exact resume: %s is visible to %d of %d ranks; put it where every rank reads it (rank 0 wrote it), or pass "
                             "--no-resume-optimizer for the reference's network-only resume" % (optim_path(resume_epoch - 1), int(round(seen.item())), world))
    if have:
        bundle = torch.load(optim_path(resume_epoch - 1), map_location='cpu', weights_only=False)
        if int(bundle['n_ave_grad']) != nAveGrad or int(bundle['n_items']) != len(trainset):
            raise SystemExit("exact resume: %s was written with nAveGrad %d over %d frames, this run has %d / %d"
                             % (optim_path(resume_epoch - 1), bundle['n_ave_grad'], bundle['n_items'], nAveGrad, len(trainset)))
        net.load_state_dict(bundle['net'])
        if hasattr(net, 'invalidate_packed_weights'):
            net.invalidate_packed_weights()
        optimizer.load_state_dict(bundle['optimizer'])
        start_iteration = int(bundle['next_iteration'])
        print("Exact resume from %s: continuing at global iteration %d (epoch %d, position %d)"
              % (optim_path(resume_epoch - 1), start_iteration, start_iteration // len(trainset), start_iteration % len(trainset)))
    first_epoch = resume_epoch if start_iteration is None else start_iteration // len(trainset)
    sched = StepSchedule(len(trainset), nAveGrad, first_epoch, nEpochs, start_iteration=start_iteration)
    loop = TrainLoop(net, optimizer, mode='parent', n_ave_grad=nAveGrad, n_epochs=nEpochs, reducer=reducer, local_ave=local_ave,
                     loss_fn=loss_fn, max_steps=sched.total_steps)
    pending, started = [], {}          # epochs whose frames this rank has finished but whose statistics are not exchanged yet
    loop.validation = {}               # epoch -> ([5 loss sums over the whole validation set], frames), filled as the sums are exchanged
    if bundle is not None and rank == 0:
        # statistics of the iterations the window of the checkpoint had already taken from later epochs (summed over the ranks when saved)
        for e, (vals, cnt) in bundle['partial_stats'].items():
            loop._running[int(e)] = [torch.tensor(v, device=device, dtype=torch.float32) for v in vals]
            loop.counts[int(e)] = int(cnt)

    def reduce_sums(values):
        """sum over the ranks of a list of floats (float64: counts and loss sums stay exact) -- one small collective"""
        if reducer is None:
            return values
        t = torch.tensor(values, device=device, dtype=torch.float64)
        if reducer.comm is not None:       # OSVOS_DP_BACKEND=abi: osvos_comm_allreduce_f64
            reducer.comm.all_reduce(t)
        else:
            import torch.distributed as dist
            dist.all_reduce(t)
        return t.tolist()

    def validate(e):
        """The validation pass of epoch e (train_parent.py:179-205), SHARDED over the ranks (frames r, r + W, ...) and run where close_epochs
        runs: right after the optimizer step that closes the window holding the epoch's last iteration -- the one place near the end of an
        epoch where every rank holds the SAME weights and stands at the same point of its collective sequence (at the end of its own share
        of the epoch a rank may already have taken the next step, or not yet the last one: with 2079 frames and nAveGrad 10 a window
        straddles every epoch boundary).  Up to nAveGrad - 1 iterations later than the reference's; single-process runs use the same rule."""
        cbce = loss_fn
        if cbce is None:
            from layers.osvos_layers import class_balanced_cross_entropy_loss as cbce
        with torch.no_grad():
            tot, n = [0.0] * 5, 0
            for k in range(rank, len(testset), world):
                sample = testset[k]
                outputs = net.forward(sample['image'].to(device))
                for i, o in enumerate(outputs):
                    tot[i] += float(cbce(o, sample['gt'].to(device), size_average=False).item())
                n += 1
        flat = reduce_sums(tot + [float(n)])
        tot, n = flat[:5], int(round(flat[5]))
        loop.validation[e] = (tot, n)
        if rank == 0:
            for l, x in enumerate(tot):
                print('***Testing (epoch %d, %d frames) *** Loss %d: %f' % (e, n, l, x / max(1, n)))

    def close_epochs(epochs):
        """Exchange + print the statistics of `epochs` and run their validation passes: called by every rank right after the same gradient
        collective (or after the last epoch), so the small all-reduces here can never pair with another rank's gradient all-reduce."""
        for e in epochs:
            pending.remove(e)
            running, count = loop.pop_running(e), loop.pop_count(e)
            flat = reduce_sums(running + [float(count)])
            running, count = flat[:-1], int(round(flat[-1]))
            if rank == 0:
                print('[Epoch: %d, numImages: %5d]' % (e, count))
                for l, v in enumerate(running):
                    print('Loss %d: %f' % (l, v / max(1, count)))
                print("Execution time: " + str(timeit.default_timer() - started[e]))
            if testset is not None and e % nTestInterval == (nTestInterval - 1):
                validate(e)
            if args.save_optimizer and (e % snapshot) == snapshot - 1 and e != 0:
                save_bundle(e)

    def save_bundle(e):
        """network + optimizer + position, right after the optimizer step that closes epoch e (or after the last iteration of the run): every
        rank holds the same weights, no gradient is half accumulated (TrainLoop.ave == 0 unless the run ended in a partial window, whose
        gradients are dropped like the reference's), the ranks stand at the same point of their collective sequences."""
        # What the closing window already took from later epochs, summed over the ranks so that rank 0 can carry it.  WHICH epochs those are
        # comes from the schedule, not from this rank's own counters: a window that takes fewer than `world` iterations from the next epoch
        # gives some ranks none of them, and a rank that skipped the all-reduce here would pair its NEXT collective (a gradient all-reduce)
        # with the other ranks' statistics exchange.  Every rank makes the same call with the same length; ranks without a
        # share contribute zeros.
        nxt = sched.next_iteration(loop.steps)
        later = [x for x in range(e + 1, nEpochs) if x * len(trainset) < nxt]
        flat = []
        for x in later:
            have = x in loop.counts
            flat += ([float(r.item()) for r in loop._running[x]] if have else [0.0] * 5) + [float(loop.counts[x]) if have else 0.0]
        flat = reduce_sums(flat) if flat else flat
        if rank == 0:
            partial = {x: (flat[6 * i:6 * i + 5], int(round(flat[6 * i + 5]))) for i, x in enumerate(later)}
            torch.save({'net': {k: v.detach().cpu() for k, v in net.state_dict().items()}, 'optimizer': optimizer.state_dict(),
                        'next_iteration': sched.next_iteration(loop.steps), 'n_ave_grad': nAveGrad, 'n_items': len(trainset),
                        'partial_stats': partial, 'world': world}, optim_path(e))
            bundles_written.add(e)

    print("Training Network")
    for epoch in range(first_epoch, nEpochs):
        started[epoch] = timeit.default_timer()
        pending.append(epoch)          # (before its first frame: the window that closes the PREVIOUS epoch may end inside this one)
        # one permutation per epoch, the same on every rank; rank r runs the iterations g = r (mod world) of the global stream
        plan = [(idx, g) for idx, g in epoch_plan(len(trainset), epoch, nAveGrad, rank, world, seed=args.seed) if g >= sched.start]
        for sample in epoch_samples(args, trainset, plan, device, augment):
            # train_parent.py:136-137 marks the HOST tensor as requiring grad and then moves it: the input gradient is computed and copied back
            # to the host every iteration.  Here the device tensor is the leaf: the same gradient is computed, nothing is copied back (and a
            # frame that is reused in the next epoch -- the --synthetic list -- does not grow a host-side .grad that is added to every epoch).
            inputs, gts = sample['image'].to(device), sample['gt'].to(device)
            inputs = inputs.detach().requires_grad_()
            _, stepped = loop.micro_batch(inputs, gts, epoch=epoch)      # forward, 5 losses, /= nAveGrad, backward, step every local_ave
            if stepped:
                close_epochs(sched.closed_by(loop.steps, pending))      # right after the SAME gradient collective on every rank
        if (epoch % snapshot) == snapshot - 1 and epoch != 0 and rank == 0:
            torch.save(net.state_dict(), os.path.join(save_dir, modelName + '_epoch-' + str(epoch) + '.pth'))      # (the reference's snapshot)
            # A bundle of an EARLIER run for this epoch no longer belongs to the snapshot just written: a later --resume-epoch would take its
            # weights, momentum and position over this run's .pth without a word.  With --save-optimizer this run's own bundle
            # replaces it when the window that closes the epoch completes; until then, and without the flag, there is none.
            if epoch not in bundles_written and os.path.exists(optim_path(epoch)):
                os.remove(optim_path(epoch))
                print("removed %s: written by an earlier run, it does not match the snapshot of this one" % optim_path(epoch))
    close_epochs(list(pending))        # epochs that end in the trailing partial window: every rank has left the loop, same order everywhere
    if rank == 0:
        print("optimizer steps taken: %d" % loop.steps)
    return net, loop


if __name__ == '__main__':
    main()
    sys.exit(0)

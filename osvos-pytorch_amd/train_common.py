"""Shared pieces of the two entry scripts: the SGD parameter groups and the restated inner loops of
the reference's train_online.py / train_parent.py, written as functions so the same code serves the
scripts, the benchmark and the tests.  Data-parallel operation (one process per GPU, gradient
all-reduce once per optimizer step) is layered on with ``GradientAllReducer``."""
from __future__ import division

import os

import torch
import torch.optim as optim

from .layers.osvos_layers import class_balanced_cross_entropy_loss
from .parallel import GradientAllReducer


def make_sgd(net, mode, lr=1e-8, wd=0.0002, momentum=0.9, fused=None):
    """Parameter groups of train_online.py:79-88 (mode 'online', score_dsn not optimised) or
    train_parent.py:87-103 (mode 'parent').  ``fused`` (default: on, OSVOS_FUSED_SGD=0 turns it off) selects
    ``FusedSGD`` -- one launch for all tensors, same numbers and state_dict -- over ``torch.optim.SGD``."""
    if fused is None:
        fused = os.environ.get("OSVOS_FUSED_SGD", "1") != "0"
    groups = [
        {'params': [pr[1] for pr in net.stages.named_parameters() if 'weight' in pr[0]], 'weight_decay': wd, 'initial_lr': lr},
        {'params': [pr[1] for pr in net.stages.named_parameters() if 'bias' in pr[0]], 'lr': 2 * lr, 'initial_lr': 2 * lr},
        {'params': [pr[1] for pr in net.side_prep.named_parameters() if 'weight' in pr[0]], 'weight_decay': wd, 'initial_lr': lr},
        {'params': [pr[1] for pr in net.side_prep.named_parameters() if 'bias' in pr[0]], 'lr': 2 * lr, 'initial_lr': 2 * lr},
    ]
    if mode == 'parent':
        groups += [
            {'params': [pr[1] for pr in net.score_dsn.named_parameters() if 'weight' in pr[0]], 'lr': lr / 10,
             'weight_decay': wd, 'initial_lr': lr / 10},
            {'params': [pr[1] for pr in net.score_dsn.named_parameters() if 'bias' in pr[0]], 'lr': 2 * lr / 10,
             'initial_lr': 2 * lr / 10},
        ]
    groups += [
        {'params': [pr[1] for pr in net.upscale.named_parameters() if 'weight' in pr[0]], 'lr': 0, 'initial_lr': 0},
        {'params': [pr[1] for pr in net.upscale_.named_parameters() if 'weight' in pr[0]], 'lr': 0, 'initial_lr': 0},
        {'params': net.fuse.weight, 'lr': lr / 100, 'initial_lr': lr / 100, 'weight_decay': wd},
        {'params': net.fuse.bias, 'lr': 2 * lr / 100, 'initial_lr': 2 * lr / 100},
    ]
    if fused:
        from .optim import FusedSGD
        return FusedSGD(groups, lr=lr, momentum=momentum)
    return optim.SGD(groups, lr=lr, momentum=momentum)


class TrainLoop(object):
    """One micro-batch = forward, loss, ``loss /= nAveGrad``, backward; optimizer step every
    ``nAveGrad`` micro-batches (train_online.py:116-149, train_parent.py:132-172).  The running loss
    is kept on the device and only read back when the caller asks for it (the reference's
    ``loss.item()`` every iteration is a logging artefact that stalls the GPU)."""

    def __init__(self, net, optimizer, mode='online', n_ave_grad=5, n_epochs=240, reducer=None, local_ave=None, loss_fn=None):
        """``n_ave_grad``: the loss divisor = micro-batches per optimizer step summed over ALL ranks (the reference's nAveGrad).
        ``local_ave``: micro-batches THIS rank contributes to a step (default: all of them); the step -- all-reduce, SGD update,
        one-memset zeroing of the flat gradient arena -- fires after that many local backwards.  The counter persists across
        epochs like the reference's ``aveGrad`` (train_parent.py:128,163-172).  ``loss_fn``: the class-balanced BCE (default: the
        HIP kernel; the CPU tests of the data-parallel bookkeeping pass the oracle's)."""
        self.net, self.opt, self.mode = net, optimizer, mode
        self.n_ave_grad, self.n_epochs = n_ave_grad, n_epochs
        self.local_ave = int(local_ave) if local_ave else n_ave_grad
        self.loss_fn = loss_fn or class_balanced_cross_entropy_loss
        self.reducer = reducer
        self.ave = 0
        self.steps = 0
        if hasattr(net, 'set_inplace_grad_accumulation'):
            net.set_inplace_grad_accumulation(True)      # every backward of this loop is loss.backward()
        dev = next(net.parameters()).device
        self.running = [torch.zeros((), device=dev) for _ in range(5 if mode == 'parent' else 1)]

    def micro_batch(self, inputs, gts, epoch=0):
        outputs = self.net.forward(inputs)
        if self.mode == 'online':
            loss = self.loss_fn(outputs[-1], gts, size_average=False)
            self.running[0] += loss.detach()
        else:
            losses = [self.loss_fn(o, gts, size_average=False) for o in outputs]
            for r, l in zip(self.running, losses):
                r += l.detach()
            loss = (1 - epoch / self.n_epochs) * sum(losses[:-1]) + losses[-1]
        loss /= self.n_ave_grad
        if self.reducer is not None and (self.ave + 1) % self.local_ave == 0:
            self.reducer.arm()          # last micro-batch of the step: reduce each gradient group as soon as its backward is done
        loss.backward()
        self.ave += 1
        stepped = False
        if self.ave % self.local_ave == 0:
            if self.reducer is not None:
                self.reducer.all_reduce()
            self.opt.step()
            if self.reducer is not None:
                self.reducer.zero_grads()          # the flat gradient arena stays attached: one memset
            else:
                self.opt.zero_grad()
            self.ave = 0
            self.steps += 1
            stepped = True
        return loss, stepped

    def pop_running(self):
        vals = [float(r.item()) for r in self.running]
        for r in self.running:
            r.zero_()
        return vals


def check_world_divides(n_ave_grad, world):
    """Data-parallel accumulation reproduces the single-process gradient only when every rank contributes the same number
    of micro-batches to every optimizer step, i.e. when the world size divides nAveGrad.  Anything else would
    silently train on a different effective batch; refuse it and say what works."""
    if world < 1 or n_ave_grad % world != 0:
        up = -(-n_ave_grad // world) * world
        raise ValueError("nAveGrad = %d is not a multiple of the world size %d: each optimizer step would sum %d micro-batches scaled by "
                         "1/%d.  Use --n-ave-grad %d (or a world size among %s)."
                         % (n_ave_grad, world, (n_ave_grad // world) * world, n_ave_grad, up,
                            [w for w in range(1, n_ave_grad + 1) if n_ave_grad % w == 0]))
    return n_ave_grad // world


def epoch_plan(n_items, epoch, n_ave_grad, rank=0, world=1, seed=0, shuffle=True):
    """Which frames THIS rank runs in `epoch`, in order: a list of (dataset index, global iteration number).

    Every rank derives the SAME permutation of range(n_items) for an epoch (generator seeded with (seed, epoch)), so across
    ranks each frame is visited exactly once per epoch and nothing is decoded twice.  The global micro-batch stream
    g = epoch * n_items + position is cut into optimizer steps of n_ave_grad consecutive iterations -- across epoch
    boundaries, like the reference's persistent ``aveGrad`` counter (train_parent.py:128,163-172) -- and inside a step rank r
    takes slots r, r + W, ...: every rank contributes n_ave_grad / W micro-batches to every step, all ranks enter the same
    number of all-reduces, and W x (n_ave_grad / W) micro-batches of per-frame class-balanced losses sum to the
    single-process gradient of the same stream (up to summation order)."""
    check_world_divides(n_ave_grad, world)
    if shuffle:
        g = torch.Generator().manual_seed(int(seed) * 1000003 + int(epoch))
        perm = torch.randperm(n_items, generator=g).tolist()
    else:
        perm = list(range(n_items))
    g0 = epoch * n_items
    # slot (g mod n_ave_grad) mod W == g mod W because W divides n_ave_grad: a plain round-robin over the global stream, so
    # ANY n_ave_grad consecutive iterations (also a window that starts at a resume point) hold n_ave_grad / W of every rank
    return [(idx, g0 + j) for j, idx in enumerate(perm) if (g0 + j) % world == rank]


def init_distributed():
    """One process per GPU (torchrun / torch.distributed.run).  Returns (rank, world, device)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
        device = torch.device('cuda', local)
    else:
        device = torch.device('cpu')
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl' if torch.cuda.is_available() else 'gloo')
    return rank, world, device


def make_reducer(net, world, average=False):
    """The gradient exchange of the data-parallel loop: torch.distributed ('nccl' = RCCL) by default, RCCL through the library's own C ABI
    with OSVOS_DP_BACKEND=abi (osvos_comm_*: no process group needed for the gradients; the chunked overlap is cheap there)."""
    if world <= 1:
        return None
    comm = None
    if os.environ.get('OSVOS_DP_BACKEND', 'torch') == 'abi' and torch.cuda.is_available():
        from .parallel import AbiCommunicator
        comm = AbiCommunicator(int(os.environ.get('RANK', '0')), world, torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0'))))
    return GradientAllReducer(net, average=average, comm=comm)
